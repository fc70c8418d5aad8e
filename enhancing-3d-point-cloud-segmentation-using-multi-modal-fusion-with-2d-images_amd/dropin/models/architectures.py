"""KPFCNN segmentation network, baseline variant (reference KPConv-PyTorch/models/architectures.py:
p2p_fitting_regularizer :25-58, KPFCNN :189-394). Module / parameter names match the reference
(encoder_blocks.N.*, decoder_blocks.N.*, head_mlp.*, head_softmax.*) for state-dict compatibility."""
import os

import numpy as np
import torch
import torch.nn as nn

try:
    from .blocks import KPConv, UnaryBlock, NearestUpsampleBlock, block_decider
    from .._native import ops as _ops
except ImportError:
    from models.blocks import KPConv, UnaryBlock, NearestUpsampleBlock, block_decider
    from _native import ops as _ops


_FUSED_REG = os.environ.get("MVK_FUSED_REGULARIZER", "1") == "1"   # development switch: 0 = the tensor-op form below
_FUSED_LOSS = os.environ.get("MVK_FUSED_LOSS", "1") == "1"         # development switch: 0 = torch.nn.CrossEntropyLoss
_FUSED_UPSAMPLE = os.environ.get("MVK_FUSED_UPSAMPLE", "1") == "1"  # development switch: 0 = closest_pool, then torch.cat
_FUSED_UPSAMPLE_LINEAR = os.environ.get("MVK_FUSED_UPSAMPLE_LINEAR", "1") == "1"   # ... and the unary layer behind them: one backward product


def p2p_fitting_regularizer(net):
    """Deformable-kernel regulariser (architectures.py:25-58): fitting loss = mean |min_d2| / extent^2
    (kernel point -> closest input point) and repulsive loss between the deformed kernel points of a
    point (the "other" points detached, :52). Same arithmetic as the reference's per-kernel-point
    Python loop, evaluated for all K points at once (one [N,K,K] distance tensor instead of 15 x 8 small
    kernels per deformable layer); in capacity-padded mode the means run over the valid rows only.
    On the GPU each layer's term and both its gradients come from ONE launch (ops.deform_regularizer)."""
    total = 0
    deform = [m for m in net.modules() if isinstance(m, KPConv) and m.deformable]
    if _FUSED_REG and deform and all(m.min_d2.is_cuda for m in deform):
        # one launch per layer each way (csrc/deform.hip), all layers one autograd node accumulating into one scalar
        term = _ops.deform_regularizer_all([(m.min_d2, m.deformed_KP, m.KP_extent, net.repulse_extent,
                                             net.deform_fitting_power, _ops.row_count_for(m.min_d2.shape[0]))
                                            for m in deform])
        return term if term is not None else total
    for m in deform:
        rows = m.min_d2.shape[0]
        n_valid = _ops.row_count_for(rows)
        if n_valid is None:
            mask, denom = None, float(rows)
        else:
            mask = (torch.arange(rows, device=m.min_d2.device) < n_valid).to(m.min_d2.dtype).unsqueeze(1)
            denom = n_valid.to(m.min_d2.dtype)

        def mean_rows(v):                      # nn.L1Loss(v, 0) = mean |v| over rows x columns (v >= 0 here)
            v = v.abs()
            if mask is not None:
                v = v * mask
            return v.sum() / (denom * v.shape[1])

        # fitting: squared distance to the closest input point, normalised by the extent (:35-38)
        fitting_loss = mean_rows(m.min_d2 / (m.KP_extent ** 2))
        # repulsion (:44-56): d[n,i,j] = |KP_i - sg(KP_j)|, j != i
        KP_locs = m.deformed_KP / m.KP_extent
        K = KP_locs.shape[1]
        diff = KP_locs.unsqueeze(2) - KP_locs.detach().unsqueeze(1)               # [N,K,K,3]
        eye = torch.eye(K, device=diff.device, dtype=diff.dtype)
        d2 = torch.sum(diff ** 2, dim=3) + eye                                     # keep sqrt'(0) off the diagonal
        dist = torch.sqrt(d2)
        rep = torch.clamp_max(dist - net.repulse_extent, max=0.0) ** 2 * (1 - eye)
        rep_loss = rep.sum(dim=2)                                                  # [N,K]: sum over the other points
        repulsive_loss = mean_rows(rep_loss) * rep_loss.shape[1] / net.K
        total = total + net.deform_fitting_power * (2 * fitting_loss + repulsive_loss)
    return total


def build_encoder(config, in_dim):
    """Encoder block list + skip bookkeeping shared by every KPFCNN variant
    (architectures.py:207-252). Returns (blocks, skips, skip_dims, in_dim, out_dim, layer, r)."""
    layer = 0
    r = config.first_subsampling_dl * config.conv_radius
    out_dim = config.first_features_dim
    blocks, skips, skip_dims = nn.ModuleList(), [], []
    for block_i, block in enumerate(config.architecture):
        if ('equivariant' in block) and (not out_dim % 3 == 0):
            raise ValueError('Equivariant block but features dimension is not a factor of 3')
        if np.any([tmp in block for tmp in ['pool', 'strided', 'upsample', 'global']]):
            skips.append(block_i)
            skip_dims.append(in_dim)
        if 'upsample' in block:
            break
        blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
        in_dim = out_dim // 2 if 'simple' in block else out_dim
        if 'pool' in block or 'strided' in block:
            layer += 1
            r *= 2
            out_dim *= 2
    return blocks, skips, skip_dims, in_dim, out_dim, layer, r


def build_decoder(config, in_dim, out_dim, layer, r, skip_dims):
    """Decoder block list (architectures.py:254-291). Returns (blocks, concats, out_dim)."""
    blocks, concats = nn.ModuleList(), []
    start_i = 0
    for block_i, block in enumerate(config.architecture):
        if 'upsample' in block:
            start_i = block_i
            break
    for block_i, block in enumerate(config.architecture[start_i:]):
        if block_i > 0 and 'upsample' in config.architecture[start_i + block_i - 1]:
            in_dim += skip_dims[layer]
            concats.append(block_i)
        blocks.append(block_decider(block, r, in_dim, out_dim, layer, config))
        in_dim = out_dim
        if 'upsample' in block:
            layer -= 1
            r *= 0.5
            out_dim = out_dim // 2
    return blocks, concats, out_dim


class _SegmentationLossMixin:
    """loss / accuracy shared by all variants (architectures.py:345-394)."""

    def _init_losses(self, config, lbl_values, ign_lbls):
        self.valid_labels = np.sort([c for c in lbl_values if c not in ign_lbls])
        if len(config.class_w) > 0:
            class_w = torch.from_numpy(np.array(config.class_w, dtype=np.float32))
            self.criterion = torch.nn.CrossEntropyLoss(weight=class_w, ignore_index=-1)
        else:
            self.criterion = torch.nn.CrossEntropyLoss(ignore_index=-1)
        self.deform_fitting_mode = config.deform_fitting_mode
        self.deform_fitting_power = config.deform_fitting_power
        self.deform_lr_factor = config.deform_lr_factor
        self.repulse_extent = config.repulse_extent
        self.output_loss = 0
        self.reg_loss = 0
        self.l1 = nn.L1Loss()

    def _targets(self, labels):
        """Ignored labels -> -1, the others -> [0, C-1] (architectures.py:352-355) through a lookup table:
        three launches whatever the number of classes, no boolean-mask assignment, no host sync
        (hipGraph capturable). Any value that is not a valid label maps to -1, like the reference's loop."""
        lut = self._label_table(labels.device)
        idx = labels.long().clamp(-1, lut.numel() - 2) + 1
        return lut[idx].to(labels.dtype)

    def _label_table(self, device, dtype=torch.int64):
        """lut[l + 1] = class index of raw label l, -1 for ignored ones (slot 0: negative labels, last slot: labels
        above the largest valid one)."""
        key = "_target_lut" if dtype == torch.int64 else "_target_lut32"
        lut = getattr(self, key, None)
        if lut is None or lut.device != device:
            top = int(max(self.valid_labels)) if len(self.valid_labels) else 0
            table = np.full(top + 3, -1, dtype=np.int64)            # slot 0: negatives, slot top+2: above the range
            for i, c in enumerate(self.valid_labels):
                table[int(c) + 1] = i
            lut = torch.from_numpy(table).to(device=device, dtype=dtype)
            setattr(self, key, lut)
        return lut

    def loss(self, outputs, labels):
        if _FUSED_LOSS and outputs.is_cuda and outputs.dtype == torch.float32:
            # label renumbering + weighted cross entropy + mean as one autograd node (csrc/loss.hip)
            self.output_loss = _ops.cross_entropy_lut(outputs, labels, self._label_table(outputs.device, torch.int32),
                                                      self.criterion.weight)
        else:
            target = self._targets(labels)
            outputs = torch.transpose(outputs, 0, 1).unsqueeze(0)
            self.output_loss = self.criterion(outputs, target.unsqueeze(0))
        if self.deform_fitting_mode == 'point2point':
            self.reg_loss = p2p_fitting_regularizer(self)
        elif self.deform_fitting_mode == 'point2plane':
            raise ValueError('point2plane fitting mode not implemented yet.')
        else:
            raise ValueError('Unknown fitting mode: ' + self.deform_fitting_mode)
        if isinstance(self.reg_loss, (int, float)) and self.reg_loss == 0:
            return self.output_loss            # rigid network: `+ 0` would be one more launch for the same value
        return self.output_loss + self.reg_loss

    def accuracy(self, outputs, labels):
        target = self._targets(labels)
        predicted = torch.argmax(outputs.data, dim=1)
        return (predicted == target).sum().item() / target.size(0)


def run_encoder_decoder(net, x, batch, encoder=None):
    """Encoder / decoder walk of KPFCNN.forward (architectures.py:322-343). When ``net.backward_cut`` holds an
    encoder block index (set ONLY by a harness that then calls dp.two_stage_backward instead of loss.backward()),
    the autograd graph is severed there: the activations that cross the cut -- the input of that block and the skip
    tensors recorded before it -- are replaced by detached leaves for everything downstream, and
    ``net.cut_tensors = (originals, leaves)`` lets the second stage continue from the leaves' gradients."""
    skip_x = []
    cut = getattr(net, "backward_cut", None) if encoder is None else None
    if cut is not None:
        net.cut_tensors = ([], [])
    for block_i, block_op in enumerate(encoder if encoder is not None else net.encoder_blocks):
        if cut is not None and block_i == cut and torch.is_grad_enabled():
            orig = [x] + list(skip_x)
            leaves = [t.detach().requires_grad_() for t in orig]
            net.cut_tensors = (orig, leaves)
            x, skip_x = leaves[0], leaves[1:]
        if block_i in net.encoder_skips:
            skip_x.append(x)
        x = block_op(x, batch)
        alias = getattr(block_op, 'skip_alias', None)
        if alias is not None:
            block_op.skip_alias = None
            if block_i in net.encoder_skips and torch.is_grad_enabled():
                skip_x[-1] = alias      # same values; its gradient reaches the block input through the block's own nodes
    return run_decoder(net, x, skip_x, batch)


def run_decoder(net, x, skip_x, batch):
    """Decoder walk (architectures.py:332-337): nearest upsampling, concatenation with the skip features, unary."""
    joined, pending = False, None
    blocks = list(net.decoder_blocks)
    for block_i, block_op in enumerate(blocks):
        if pending is not None:         # the unary layer behind a fused upsampling + concatenation takes all three
            x = block_op.forward_upsampled(*pending)
            pending = None
            continue
        if block_i in net.decoder_concats and not joined:
            x = torch.cat([x, skip_x.pop()], dim=1)
        joined = False
        if _FUSED_UPSAMPLE and isinstance(block_op, NearestUpsampleBlock) and (block_i + 1) in net.decoder_concats \
                and x.is_cuda and block_i not in net.decoder_concats:
            inds, skip = batch.upsamples[block_op.layer_ind - 1], skip_x.pop()
            nxt = blocks[block_i + 1] if block_i + 1 < len(blocks) else None
            if _FUSED_UPSAMPLE_LINEAR and isinstance(nxt, UnaryBlock) and torch.is_grad_enabled() and net.training:
                pending = (x, inds, skip)
                continue
            # nearest upsampling + the concatenation that follows it (:334-335) as one launch
            x = _ops.upsample_cat(x, inds, skip)
            joined = True
            continue
        x = block_op(x, batch)
    return x


class KPFCNN(_SegmentationLossMixin, nn.Module):
    """3D-only KPFCNN (train_ScanNet_baseline.py)."""

    def __init__(self, config, lbl_values, ign_lbls):
        super(KPFCNN, self).__init__()
        self.K = config.num_kernel_points
        self.C = len(lbl_values) - len(ign_lbls)
        (self.encoder_blocks, self.encoder_skips, self.encoder_skip_dims,
         in_dim, out_dim, layer, r) = build_encoder(config, config.in_features_dim)
        self.decoder_blocks, self.decoder_concats, out_dim = build_decoder(
            config, in_dim, out_dim, layer, r, self.encoder_skip_dims)
        self.head_mlp = UnaryBlock(out_dim, config.first_features_dim, False, 0)
        self.head_softmax = UnaryBlock(config.first_features_dim, self.C, False, 0)
        self._init_losses(config, lbl_values, ign_lbls)

    def forward(self, batch, config):
        x = batch.features.detach()          # (`.clone().detach()` in the reference; nothing below writes x in place)
        x = run_encoder_decoder(self, x, batch)
        return self.head_softmax(self.head_mlp(x, batch), batch)
