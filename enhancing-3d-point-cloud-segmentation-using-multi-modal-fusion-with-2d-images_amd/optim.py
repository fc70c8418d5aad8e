"""Optimiser tail of the reference's training step on one HIP launch.

The reference's trainer ends every step with ``torch.nn.utils.clip_grad_value_(net.parameters(),
config.grad_clip_norm)`` and ``optimizer.step()`` of a ``torch.optim.SGD`` with momentum and weight decay built
from two parameter groups -- parameters whose name contains 'offset' get ``lr * deform_lr_factor``
(KPConv-PyTorch/utils/trainer.py:72-79, 190-195). As library calls that is ~50 small launches per step (one
clamp per tensor plus the multi-tensor chunks); ``FusedClipSGD`` walks all tensors with a single grid
(csrc/optim.hip, mvk_sgd_clip_step). Same arithmetic as torch.optim.SGD (dampening 0, no Nesterov)."""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check

_REC = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("n", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
assert _REC.itemsize == 40


class FusedClipSGD:
    """groups: list of {"params": [...], optional "lr", "weight_decay"} like torch.optim.SGD.
    clip_value: the bound of clip_grad_value_ (None / inf: no clipping). clip_in_place also leaves the clamped
    values in ``.grad`` like clip_grad_value_ does (costs one more store per parameter; off by default)."""

    def __init__(self, groups, lr, momentum=0.0, weight_decay=0.0, clip_value=None, clip_in_place=False):
        if isinstance(groups, (list, tuple)) and groups and isinstance(groups[0], torch.Tensor):
            groups = [{"params": list(groups)}]
        self.param_groups = []
        for g in groups:
            ps = [p for p in g["params"] if p.requires_grad]
            self.param_groups.append({"params": ps, "lr": float(g.get("lr", lr)),
                                      "weight_decay": float(g.get("weight_decay", weight_decay))})
        self.momentum = float(momentum)
        # trainer.py:191: `if self.config.grad_clip_norm > 0: clip_grad_value_(...)` -- a bound <= 0 means NO clipping
        self.clip = float("inf") if (clip_value is None or float(clip_value) <= 0) else float(clip_value)
        self.clip_in_place = bool(clip_in_place)
        self.params = [p for g in self.param_groups for p in g["params"]]
        for p in self.params:
            if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("FusedClipSGD needs contiguous float32 parameters resident in HBM (no CPU path)")
        # one flat zero-initialised momentum buffer (m = momentum * 0 + d on the first step = torch's clone(d))
        self._flat_m = torch.zeros(sum(p.numel() for p in self.params), dtype=torch.float32,
                                   device=self.params[0].device) if self.params else None
        self.state, off = {}, 0
        for p in self.params:
            self.state[p] = {"momentum_buffer": self._flat_m[off:off + p.numel()].view_as(p)}
            off += p.numel()
        self._chunk = lib().mvk_sgd_chunk_elems()
        self._tables = {}                # eager steps: (pointers, rates) of a step -> the slot that holds its table
        self._cur = None
        # Table slots (pinned host staging + device copy) are allocated HERE: pinning memory is not allowed while a
        # stream is capturing. Eager steps rotate through two slots (guarded by events); every graph capture takes
        # a slot of its own, because the captured memcpy node re-reads the pinned bytes at each replay.
        self._max_chunks = sum((p.numel() + self._chunk - 1) // self._chunk for p in self.params)
        self._free = [self._new_slot() for _ in range(12)] if self.params else []
        self._eager = []
        self._captured = []

    def _new_slot(self):
        dev = self.params[0].device
        nb = max(len(self.params), 1) * _REC.itemsize
        return {"event": None, "n": 0,
                "host": torch.empty(nb, dtype=torch.uint8).pin_memory(),
                "dev": torch.empty(nb, dtype=torch.uint8, device=dev),
                "chunks_host": torch.empty(max(self._max_chunks, 1) * 2, dtype=torch.int32).pin_memory(),
                "chunks_dev": torch.empty(max(self._max_chunks, 1) * 2, dtype=torch.int32, device=dev)}

    def zero_grad(self, set_to_none=True):
        for p in self.params:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()

    def set_lr(self, lr, group=None):
        for i, g in enumerate(self.param_groups):
            if group is None or group == i:
                g["lr"] = float(lr)
        self.sync_hyperparameters()

    def sync_hyperparameters(self):
        """Call after changing ``param_groups[i]['lr']`` / ``['weight_decay']`` in place (the reference's trainer does
        at every epoch end, trainer.py:239-241). Eager steps pick the new values up by themselves (the table key holds
        them); the tables of CAPTURED steps are rewritten here: a captured memcpy node re-reads its pinned host bytes
        at every replay, so the next replay of every graph runs with the new rates. Returns the number of captured
        tables rewritten."""
        self._tables = {}
        group_of = {id(p): g for g in self.param_groups for p in g["params"]}
        by_ptr = {p.data_ptr(): group_of[id(p)] for p in self.params}
        n = 0
        for slot in self._captured:
            k = slot.get("records", 0)
            rec = slot["host"].numpy()[:k * _REC.itemsize].view(_REC)
            for i in range(k):
                g = by_ptr.get(int(rec[i]["p"]))
                if g is not None:
                    rec[i]["lr"], rec[i]["wd"] = g["lr"], g["weight_decay"]
            n += 1
        return n

    def state_dict(self):
        """torch.optim.SGD's layout (trainer.py:251 saves ``optimizer.state_dict()``): ``state`` maps the parameter's
        position to ``{'momentum_buffer': tensor}``, ``param_groups`` hold the hyper-parameters and position lists."""
        pos = {id(p): i for i, p in enumerate(self.params)}
        groups = [{"lr": g["lr"], "momentum": self.momentum, "dampening": 0, "weight_decay": g["weight_decay"],
                   "nesterov": False, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                   "params": [pos[id(p)] for p in g["params"]]} for g in self.param_groups]
        return {"state": {i: {"momentum_buffer": self.state[p]["momentum_buffer"].clone()} for i, p in enumerate(self.params)},
                "param_groups": groups}

    def load_state_dict(self, sd):
        """Accepts a state dict of this class or of a torch.optim.SGD over the same parameter order (trainer.py:101
        restores ``optimizer_state_dict`` from a checkpoint). Momentum buffers are copied INTO the flat buffer the
        captured graphs point at; learning rates / weight decay are taken over and pushed to captured tables."""
        groups = sd["param_groups"]
        if len(groups) != len(self.param_groups) or any(len(a["params"]) != len(b["params"])
                                                        for a, b in zip(groups, self.param_groups)):
            raise ValueError("loaded state dict has different parameter groups")
        for mine, theirs in zip(self.param_groups, groups):
            mine["lr"], mine["weight_decay"] = float(theirs["lr"]), float(theirs.get("weight_decay", mine["weight_decay"]))
        if groups and "momentum" in groups[0]:
            self.momentum = float(groups[0]["momentum"])
        with torch.no_grad():
            for i, p in enumerate(self.params):
                st = sd["state"].get(i, sd["state"].get(str(i)))
                buf = None if st is None else st.get("momentum_buffer")
                if buf is None:
                    self.state[p]["momentum_buffer"].zero_()       # torch: no buffer yet == first step == zeros here
                else:
                    self.state[p]["momentum_buffer"].copy_(buf.to(p.device, torch.float32).view_as(p))
        self.sync_hyperparameters()

    def _build(self, items):
        dev = self.params[0].device
        rec = np.zeros(len(items), dtype=_REC)
        chunks = []
        for i, (p, g, lr, wd) in enumerate(items):
            rec[i] = (p.data_ptr(), g.data_ptr(), self.state[p]["momentum_buffer"].data_ptr(), p.numel(), lr, wd)
            chunks.append(np.stack([np.full((p.numel() + self._chunk - 1) // self._chunk, i, np.int32),
                                    np.arange((p.numel() + self._chunk - 1) // self._chunk, dtype=np.int32)], 1))
        chunks = np.concatenate(chunks, 0) if chunks else np.zeros((0, 2), np.int32)
        capturing = torch.cuda.is_current_stream_capturing()
        if capturing:
            if not self._free:
                raise RuntimeError("FusedClipSGD: out of pre-pinned table slots (more than 4 graph captures); "
                                   "create the optimiser with more slots before capturing")
            slot = self._free.pop()
            self._captured.append(slot)
        else:
            while len(self._eager) < 4 and len(self._free) > 8:
                self._eager.append(self._free.pop())
            if not self._eager:
                self._eager.append(self._new_slot())
            slot = self._eager.pop(0)
            self._eager.append(slot)
            if slot["event"] is not None:
                slot["event"].synchronize()      # its previous table copy has been consumed
        slot["host"].numpy()[:rec.nbytes] = rec.view(np.uint8).reshape(-1)
        slot["chunks_host"].numpy()[:chunks.size] = chunks.reshape(-1)
        slot["dev"].copy_(slot["host"], non_blocking=True)          # under capture: a memcpy node replayed with the graph
        slot["chunks_dev"].copy_(slot["chunks_host"], non_blocking=True)
        slot["n"] = int(chunks.shape[0])
        slot["records"] = len(items)
        return slot

    @torch.no_grad()
    def step(self, only=None):
        """only: an iterable of parameters -- step just those (bench.py steps the parameters above the backward cut on
        a side stream while the rest of the backward is still running, the others at the end)."""
        items = []
        subset = None if only is None else {id(p) for p in only}
        for g in self.param_groups:
            for p in g["params"]:
                if p.grad is None or (subset is not None and id(p) not in subset):
                    continue
                gr = p.grad
                if gr.dtype != torch.float32 or not gr.is_contiguous():
                    gr = gr.float().contiguous()
                    p.grad = gr
                items.append((p, gr, g["lr"], g["weight_decay"]))
        if not items:
            return
        key = tuple((p.data_ptr(), g.data_ptr(), lr, wd) for p, g, lr, wd in items)
        capturing = torch.cuda.is_current_stream_capturing()
        cached = self._tables.get(key) if not capturing else None
        if cached is None:
            cached = self._build(items)
            # a captured slot's device table is only filled when its graph replays: an eager step must never reuse it
            if not capturing:
                self._tables = {k: v for k, v in self._tables.items() if v is not cached}    # a rotated slot lost its old table
                self._tables[key] = cached
        t = self._cur = cached
        stream = torch.cuda.current_stream()
        check(lib().mvk_sgd_clip_step(C.c_void_p(t["dev"].data_ptr()), C.c_void_p(t["chunks_dev"].data_ptr()), t["n"],
                                      self.clip if np.isfinite(self.clip) else 3.0e38, self.momentum,
                                      int(self.clip_in_place), C.c_void_p(stream.cuda_stream)))
        if not capturing:
            ev = torch.cuda.Event()
            ev.record(stream)
            t["event"] = ev
