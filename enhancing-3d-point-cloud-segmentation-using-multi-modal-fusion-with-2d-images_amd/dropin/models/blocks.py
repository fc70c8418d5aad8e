"""KPConv operator and network blocks with the reference's API
(KPConv-PyTorch/models/blocks.py), running on the fused HIP kernels.

Same class names, constructor arguments, attribute names and state-dict keys as the reference
(``weights``, ``kernel_points``, ``offset_conv.*``, ``offset_bias``, ``KPConv.*``, ``batch_norm.*``,
``mlp.weight`` ...) so checkpoints and the trainer's parameter grouping (names containing
'offset', utils/trainer.py:72-73) keep working. The tensor algebra of KPConv.forward
(blocks.py:277-374) is one autograd node here (ops._KPConvFn): gather + correlation + aggregation
kernel, f32 MFMA contraction, scatter backward.
"""
import math
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn.init import kaiming_uniform_
from torch.nn.parameter import Parameter

try:
    from .._native import ops
    from ..kernels.kernel_points import load_kernels
except ImportError:  # dropin/ put on sys.path directly
    from _native import ops
    from kernels.kernel_points import load_kernels


_FUSE_ADD = os.environ.get("MVK_FUSE_ADD", "1") == "1"      # development switches for A/B timing
_HIP_BN = os.environ.get("MVK_HIP_BN", "1") == "1"          # training-mode BatchNorm on the masked HIP kernel (0: nn.BatchNorm1d)
_GEMM_STATS = os.environ.get("MVK_GEMM_STATS", "1") == "1"  # BatchNorm statistics from the producing GEMM's epilogue
_FUSE_FANOUT = os.environ.get("MVK_FUSE_FANOUT", "1") == "1"    # the two gradients of a block's input summed inside unary1's backward GEMM
_FUSED_BIAS = os.environ.get("MVK_FUSED_BIAS", "1") == "1"      # bias + LeakyReLU of the BatchNorm-less layers in one launch
_FUSED_OPERANDS = os.environ.get("MVK_FUSED_DEFORM_OPERANDS", "1") == "1"   # development switch: 0 = the tensor ops
_GEMM_PAIR = os.environ.get("MVK_GEMM_PAIR", "1") == "1"        # unary1 and the shortcut layer of a block (same input) as one GEMM launch
_BN_PAIR = os.environ.get("MVK_BN_PAIR", "1") == "1"            # a block's two independent BatchNorms (convolution, shortcut) as one launch each way
_BN_FOLD_CONV = os.environ.get("MVK_BN_FOLD_CONV", "1") == "1"  # batch_norm_conv + LeakyReLU inside unary2's operand load (needs MVK_BN_FOLD)
_ORDER_LOOKUP = os.environ.get("MVK_GATHER_ORDER", "1") != "0"   # gather work lists found by their points tensor (ops.work_order_for)


def _bn_rows(x, module, use_bn=True):
    """DEVICE row count the BatchNorm after a layer will normalise over (capacity-padded level: its valid rows;
    otherwise all rows), or None when that BatchNorm does not run on the HIP kernel."""
    if not (use_bn and module.training and x.is_cuda):
        return None
    n_valid = ops.row_count_for(x.shape[0])
    if n_valid is None and _HIP_BN:
        n_valid = ops.full_count(x.shape[0], x.device)
    return n_valid

_MFMA_LINEAR = os.environ.get("MVK_MFMA_LINEAR", "1") == "1"


# ---------------------------------------------------------------- simple functions (blocks.py:35-133)

def gather(x, idx, method=2):
    """x[idx] for row indices of any shape (blocks.py:35-66). The reference's expand+gather forms
    (methods 1, 2) exist for a faster backward of the materialised [N,H,D] gather; the fused
    kernels never materialise it, so plain indexing is all that is left to do here."""
    return x[idx]


def radius_gaussian(sq_r, sig, eps=1e-9):
    return torch.exp(-sq_r / (2 * sig ** 2 + eps))


def closest_pool(x, inds):
    """Features of the closest (first-column) neighbour; shadow index -> zeros (blocks.py:79-91)."""
    return ops.closest_pool(x, inds)


def max_pool(x, inds):
    """Max over the neighbourhood, the zero shadow row taking part (blocks.py:94-110)."""
    return ops.max_pool(x, inds)


def global_average(x, batch_lengths):
    """Per-cloud mean over the stacked point axis (blocks.py:113-133)."""
    out, i0 = [], 0
    for length in batch_lengths:
        length = int(length)
        out.append(torch.mean(x[i0:i0 + length], dim=0))
        i0 += length
    return torch.stack(out)


# ---------------------------------------------------------------- KPConv (blocks.py:143-379)

class KPConv(nn.Module):

    def __init__(self, kernel_size, p_dim, in_channels, out_channels, KP_extent, radius,
                 fixed_kernel_points='center', KP_influence='linear', aggregation_mode='sum',
                 deformable=False, modulated=False):
        super(KPConv, self).__init__()
        self.K = kernel_size
        self.p_dim = p_dim
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.radius = radius
        self.KP_extent = KP_extent
        self.fixed_kernel_points = fixed_kernel_points
        self.KP_influence = KP_influence
        self.aggregation_mode = aggregation_mode
        self.deformable = deformable
        self.modulated = modulated
        if p_dim != 3:
            raise ValueError('the HIP KPConv kernels are 3-D (p_dim == 3)')

        # running variables read by the regulariser (architectures*.py p2p_fitting_regularizer)
        self.min_d2 = None
        self.deformed_KP = None
        self.offset_features = None

        self.weights = Parameter(torch.zeros((self.K, in_channels, out_channels), dtype=torch.float32),
                                 requires_grad=True)
        if deformable:
            self.offset_dim = (self.p_dim + 1) * self.K if modulated else self.p_dim * self.K
            self.offset_conv = KPConv(self.K, self.p_dim, self.in_channels, self.offset_dim, KP_extent, radius,
                                      fixed_kernel_points=fixed_kernel_points, KP_influence=KP_influence,
                                      aggregation_mode=aggregation_mode)
            self.offset_bias = Parameter(torch.zeros(self.offset_dim, dtype=torch.float32), requires_grad=True)
        else:
            self.offset_dim = None
            self.offset_conv = None
            self.offset_bias = None
        self.reset_parameters()
        self.kernel_points = self.init_KP()

    def reset_parameters(self):
        kaiming_uniform_(self.weights, a=math.sqrt(5))
        if self.deformable:
            nn.init.zeros_(self.offset_bias)

    def init_KP(self):
        kp = load_kernels(self.radius, self.K, dimension=self.p_dim, fixed=self.fixed_kernel_points)
        return Parameter(torch.tensor(kp, dtype=torch.float32), requires_grad=False)

    def forward(self, q_pts, s_pts, neighb_inds, x, stats_n_valid=None, order=None, rev=None, rev_order=None, bn=None):
        """stats_n_valid: DEVICE row count of the BatchNorm that follows this convolution (the contraction
        then delivers its statistics, ops.bn_stats_of); order: work list of the query points for the gather
        (ops.kpconv); rev / rev_order: the transposed neighbour matrix and the support level's work list for the
        gather-form feature gradient (ops.reverse_neighbors); bn: the nn.BatchNorm1d that follows -- the contraction then
        FINISHES its statistics (ops.kpconv). None of them is part of the reference signature."""
        if self.KP_influence not in ops.INFLUENCE:
            raise ValueError('Unknown influence function type (config.KP_influence)')
        if self.aggregation_mode not in ops.AGGREGATION:
            raise ValueError("Unknown convolution mode. Should be 'closest' or 'sum'")
        offsets = modulations = None
        if self.deformable:
            # offsets from an inner rigid KPConv (blocks.py:243-266)
            raw = self.offset_conv(q_pts, s_pts, neighb_inds, x, order=order, rev=rev, rev_order=rev_order)
            if _FUSED_OPERANDS and raw.is_cuda:        # bias, scale, kernel points (and 2 sigmoid) in one launch
                self.offset_features, offsets, self.deformed_KP, modulations = ops.deform_operands(
                    raw, self.offset_bias, self.kernel_points, self.KP_extent, self.modulated)
            else:
                self.offset_features = raw + self.offset_bias
                nk = self.p_dim * self.K
                if self.modulated:
                    unscaled = self.offset_features[:, :nk].reshape(-1, self.K, self.p_dim)
                    modulations = 2 * torch.sigmoid(self.offset_features[:, nk:])
                else:
                    unscaled = self.offset_features.reshape(-1, self.K, self.p_dim)
                offsets = unscaled * self.KP_extent
                self.deformed_KP = offsets + self.kernel_points          # blocks.py:287
        y, min_d2 = ops.kpconv(q_pts, s_pts, neighb_inds, x, self.kernel_points, self.weights, self.KP_extent,
                               self.KP_influence, self.aggregation_mode, offsets, modulations,
                               stats_n_valid=stats_n_valid, order=order, rev=rev, rev_order=rev_order, bn=bn)
        if self.deformable:
            self.min_d2 = min_d2                                      # blocks.py:303
        return y

    def __repr__(self):
        return 'KPConv(radius: {:.2f}, in_feat: {:d}, out_feat: {:d})'.format(self.radius, self.in_channels,
                                                                              self.out_channels)


# ---------------------------------------------------------------- blocks (blocks.py:387-694)

_SIMPLE = {'simple', 'simple_deformable', 'simple_invariant', 'simple_equivariant', 'simple_strided',
           'simple_deformable_strided', 'simple_invariant_strided', 'simple_equivariant_strided'}
_RESNET = {'resnetb', 'resnetb_invariant', 'resnetb_equivariant', 'resnetb_deformable', 'resnetb_strided',
           'resnetb_deformable_strided', 'resnetb_equivariant_strided', 'resnetb_invariant_strided'}


def block_decider(block_name, radius, in_dim, out_dim, layer_ind, config):
    if block_name == 'unary':
        return UnaryBlock(in_dim, out_dim, config.use_batch_norm, config.batch_norm_momentum)
    if block_name in _SIMPLE:
        return SimpleBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in _RESNET:
        return ResnetBottleneckBlock(block_name, in_dim, out_dim, radius, layer_ind, config)
    if block_name in ('max_pool', 'max_pool_wide'):
        return MaxPoolBlock(layer_ind)
    if block_name == 'global_average':
        return GlobalAverageBlock()
    if block_name == 'nearest_upsample':
        return NearestUpsampleBlock(layer_ind)
    raise ValueError('Unknown block name in the architecture definition : ' + block_name)


class BatchNormBlock(nn.Module):
    """BatchNorm1d over the stacked point axis, or a learned bias (blocks.py:430-467)."""

    def __init__(self, in_dim, use_bn, bn_momentum):
        super(BatchNormBlock, self).__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.in_dim = in_dim
        if self.use_bn:
            self.batch_norm = nn.BatchNorm1d(in_dim, momentum=bn_momentum)
        else:
            self.bias = Parameter(torch.zeros(in_dim, dtype=torch.float32), requires_grad=True)

    def reset_parameters(self):
        nn.init.zeros_(self.bias)

    def forward(self, x, slope=None, addend=None):
        """slope: when given, the LeakyReLU that follows every BatchNormBlock in the reference's blocks
        is applied here (fused into the masked kernel in capacity-padded mode). addend: the shortcut of a
        residual block, added before that activation (blocks.py:649), in the same launch when masked."""
        if self.use_bn:
            n_valid = _bn_rows(x, self)
            if n_valid is not None:
                # capacity-padded level (hipGraph replay): statistics over the valid rows only
                return ops.bn_lrelu(x, n_valid, self.batch_norm, 1.0 if slope is None else slope, addend)
            # [N, C] is already BatchNorm1d's (batch, channel) layout: same statistics as the
            # reference's unsqueeze/transpose round trip (blocks.py:456-460) without the copies
            x = self.batch_norm(x)
        else:
            if addend is None and x.is_cuda and x.dim() == 2 and x.shape[1] <= 256 and _FUSED_BIAS:
                return ops.bias_lrelu(x, self.bias, 1.0 if slope is None else slope)     # one launch each way
            x = x + self.bias
        if addend is not None:
            return ops.add_lrelu(x, addend, 1.0 if slope is None else slope) if _FUSE_ADD else \
                nn.functional.leaky_relu(x + addend, 1.0 if slope is None else slope)
        return x if slope is None else nn.functional.leaky_relu(x, slope)

    def __repr__(self):
        return 'BatchNormBlock(in_feat: {:d}, momentum: {:.3f}, only_bias: {:s})'.format(
            self.in_dim, self.bn_momentum, str(not self.use_bn))


class UnaryBlock(nn.Module):

    def __init__(self, in_dim, out_dim, use_bn, bn_momentum, no_relu=False):
        super(UnaryBlock, self).__init__()
        self.bn_momentum = bn_momentum
        self.use_bn = use_bn
        self.no_relu = no_relu
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.mlp = nn.Linear(in_dim, out_dim, bias=False)
        self.batch_norm = BatchNormBlock(out_dim, self.use_bn, self.bn_momentum)
        if not no_relu:
            self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, batch=None, join=None, passthrough=False):
        """join = (shortcut, slope): finish a residual block here -- LeakyReLU_slope(BN(x W^T) + shortcut).
        passthrough: returns (output, x') where x' aliases x for the block's shortcut branch: the gradients of the two
        consumers of x are then summed inside this layer's backward GEMM (ops.linear) instead of by one more launch."""
        # nn.Linear(bias=False) = x @ W^T: on the f32 MFMA GEMM (faster than the library GEMM on these
        # tall-skinny shapes, tools/gemm_bench.py); parameters stay those of self.mlp (state-dict compatible)
        alias = x
        if (_MFMA_LINEAR and _FUSED_BIAS and x.is_cuda and not self.use_bn and join is None and not passthrough
                and x.dim() == 2 and self.out_dim <= 256):
            # BatchNorm-less layer: bias and activation leave with the GEMM's store (ops.linear_bias_lrelu)
            return ops.linear_bias_lrelu(x, self.mlp.weight, self.batch_norm.bias, 1.0 if self.no_relu else 0.1)
        if _MFMA_LINEAR and x.is_cuda:
            nv = _bn_rows(x, self, self.use_bn) if _GEMM_STATS else None
            bn = self.batch_norm.batch_norm if (self.use_bn and nv is not None) else None    # statistics finished by the product
            if passthrough and _FUSE_FANOUT and x.requires_grad:
                y, alias = ops.linear(x, self.mlp.weight, stats_n_valid=nv, passthrough=True, bn=bn)
            else:
                y = ops.linear(x, self.mlp.weight, stats_n_valid=nv, bn=bn)
        else:
            y = self.mlp(x)
        if join is not None:
            out = self.batch_norm(y, join[1], addend=join[0])
        else:
            out = self.batch_norm(y, None if self.no_relu else 0.1)
        return (out, alias) if passthrough else out

    def forward_upsampled(self, x, inds, skip):
        """forward(cat([closest_pool(x, inds), skip], 1)): the decoder's upsampling + concatenation + this layer
        (architectures.py:334-337) with the backward of all three as one product (ops.upsample_cat_linear)."""
        nv = _bn_rows(skip, self, self.use_bn) if _GEMM_STATS else None
        y = ops.upsample_cat_linear(x, inds, skip, self.mlp.weight, stats_n_valid=nv,
                                    bn=self.batch_norm.batch_norm if (self.use_bn and nv is not None) else None)
        return self.batch_norm(y, None if self.no_relu else 0.1)

    def forward_normalised(self, x_raw, n_valid, bn_in, slope_in, join=None):
        """forward(bn_lrelu(x_raw, n_valid, bn_in, slope_in), join=join) with the apply pass of bn_in folded into this
        layer's product (ops.bn_lrelu_linear), or None when x_raw does not carry finished statistics."""
        if not (_MFMA_LINEAR and self.use_bn and x_raw.is_cuda):
            return None
        nv = _bn_rows(x_raw, self, True) if _GEMM_STATS else None
        res = ops.bn_lrelu_linear(x_raw, n_valid, bn_in, slope_in, self.mlp.weight, stats_n_valid=nv,
                                  bn_out=self.batch_norm.batch_norm if nv is not None else None)
        if res is None:
            return None
        y = res[0]
        if join is not None:
            return self.batch_norm(y, join[1], addend=join[0])
        return self.batch_norm(y, None if self.no_relu else 0.1)

    def __repr__(self):
        return 'UnaryBlock(in_feat: {:d}, out_feat: {:d}, BN: {:s}, ReLU: {:s})'.format(
            self.in_dim, self.out_dim, str(self.use_bn), str(not self.no_relu))


def _conv_inputs(block_name, layer_ind, batch):
    """Which pyramid tensors a block convolves over (blocks.py:551-558, :623-630)."""
    if 'strided' in block_name:
        return batch.points[layer_ind + 1], batch.points[layer_ind], batch.pools[layer_ind]
    return batch.points[layer_ind], batch.points[layer_ind], batch.neighbors[layer_ind]


def _work_order(block_name, layer_ind, batch):
    """The cell-sorted work list of the block's query level when the batch carries one (datasets/common.py
    `orders`, not a reference attribute): the gather then walks the points region by region."""
    l = layer_ind + 1 if 'strided' in block_name else layer_ind
    orders = getattr(batch, 'orders', None)
    if orders:
        return orders[l] if l < len(orders) else None
    return ops.work_order_for(batch.points[l]) if _ORDER_LOOKUP else None


def _reverse_list(block_name, layer_ind, batch):
    """(transposed neighbour matrix, work list of the support level) of the block's convolution when the batch carries
    them (datasets/common.py `rev_neighbors` / `rev_pools`, not reference attributes), else (None, None): the feature
    gradient then runs as the atomic scatter."""
    strided = 'strided' in block_name
    revs = getattr(batch, 'rev_pools' if strided else 'rev_neighbors', None)
    if revs and layer_ind < len(revs) and revs[layer_ind] is not None:
        rev = revs[layer_ind]
    else:
        # batch containers without these attributes (the reference's flat input_list): the list registered under the
        # neighbour matrix itself when the pyramid was built in this process (ops.remember_reverse), like the work lists
        inds = (batch.pools if strided else batch.neighbors)[layer_ind]
        rev = ops.reverse_for(inds) if (_ORDER_LOOKUP and ops.REVERSE_DX and torch.is_tensor(inds)) else None
        if rev is None:
            return None, None
    orders = getattr(batch, 'orders', None)
    order = orders[layer_ind] if orders and layer_ind < len(orders) else \
        (ops.work_order_for(batch.points[layer_ind]) if _ORDER_LOOKUP else None)
    return rev, order


class SimpleBlock(nn.Module):

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(SimpleBlock, self).__init__()
        current_extent = radius * config.KP_extent / config.conv_radius
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.layer_ind = layer_ind
        self.block_name = block_name
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, in_dim, out_dim // 2,
                             current_extent, radius, fixed_kernel_points=config.fixed_kernel_points,
                             KP_influence=config.KP_influence, aggregation_mode=config.aggregation_mode,
                             deformable='deform' in block_name, modulated=config.modulated)
        self.batch_norm = BatchNormBlock(out_dim // 2, self.use_bn, self.bn_momentum)
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, x, batch):
        q_pts, s_pts, inds = _conv_inputs(self.block_name, self.layer_ind, batch)
        nv = _bn_rows(q_pts, self, self.use_bn) if _GEMM_STATS else None
        rev, rev_order = _reverse_list(self.block_name, self.layer_ind, batch) if x.requires_grad else (None, None)
        return self.batch_norm(self.KPConv(q_pts, s_pts, inds, x, stats_n_valid=nv,
                                           order=_work_order(self.block_name, self.layer_ind, batch), rev=rev,
                                           rev_order=rev_order,
                                           bn=self.batch_norm.batch_norm if (self.use_bn and nv is not None) else None), 0.1)


class ResnetBottleneckBlock(nn.Module):

    def __init__(self, block_name, in_dim, out_dim, radius, layer_ind, config):
        super(ResnetBottleneckBlock, self).__init__()
        current_extent = radius * config.KP_extent / config.conv_radius
        self.bn_momentum = config.batch_norm_momentum
        self.use_bn = config.use_batch_norm
        self.block_name = block_name
        self.layer_ind = layer_ind
        self.in_dim = in_dim
        self.out_dim = out_dim
        if in_dim != out_dim // 4:
            self.unary1 = UnaryBlock(in_dim, out_dim // 4, self.use_bn, self.bn_momentum)
        else:
            self.unary1 = nn.Identity()
        self.KPConv = KPConv(config.num_kernel_points, config.in_points_dim, out_dim // 4, out_dim // 4,
                             current_extent, radius, fixed_kernel_points=config.fixed_kernel_points,
                             KP_influence=config.KP_influence, aggregation_mode=config.aggregation_mode,
                             deformable='deform' in block_name, modulated=config.modulated)
        self.batch_norm_conv = BatchNormBlock(out_dim // 4, self.use_bn, self.bn_momentum)
        self.unary2 = UnaryBlock(out_dim // 4, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        if in_dim != out_dim:
            self.unary_shortcut = UnaryBlock(in_dim, out_dim, self.use_bn, self.bn_momentum, no_relu=True)
        else:
            self.unary_shortcut = nn.Identity()
        self.leaky_relu = nn.LeakyReLU(0.1)

    def forward(self, features, batch):
        q_pts, s_pts, inds = _conv_inputs(self.block_name, self.layer_ind, batch)
        us, ys = self.unary_shortcut, None
        if (_GEMM_PAIR and _MFMA_LINEAR and features.is_cuda and 'strided' not in self.block_name
                and isinstance(self.unary1, UnaryBlock) and isinstance(us, UnaryBlock)):
            # unary1 and the shortcut layer read the same rows: their two products as one launch (ops.linear_pair)
            nv_in = _bn_rows(features, self, self.use_bn) if _GEMM_STATS else None
            pair = ops.linear_pair(features, self.unary1.mlp.weight, us.mlp.weight, nv_in,
                                   bn0=self.unary1.batch_norm.batch_norm if (self.use_bn and nv_in is not None) else None,
                                   bn1=us.batch_norm.batch_norm if (self.use_bn and nv_in is not None) else None)
            if pair is not None:
                ys = pair[1]
                x = self.unary1.batch_norm(pair[0], 0.1)
        if ys is None:
            if isinstance(self.unary1, UnaryBlock):
                x, features = self.unary1(features, passthrough=True)  # `features` from here on: the shortcut's input
            else:
                x = self.unary1(features)
        nv = _bn_rows(q_pts, self, self.use_bn) if _GEMM_STATS else None
        rev, rev_order = _reverse_list(self.block_name, self.layer_ind, batch)
        conv = self.KPConv(q_pts, s_pts, inds, x, stats_n_valid=nv, order=_work_order(self.block_name, self.layer_ind, batch),
                           rev=rev, rev_order=rev_order,
                           bn=self.batch_norm_conv.batch_norm if (self.use_bn and nv is not None) else None)
        self.skip_alias = None
        if 'strided' in self.block_name and _FUSE_FANOUT and features.is_cuda:
            # the block's input is also the decoder's skip tensor (architectures.py:328-329): the alias handed out here
            # lets the pooling's backward accumulate onto the decoder's gradient (run_encoder_decoder picks it up)
            shortcut, self.skip_alias = ops.max_pool(features, inds, passthrough=True)
        else:
            shortcut = max_pool(features, inds) if 'strided' in self.block_name else features
        n_valid = _bn_rows(conv, self, self.use_bn) if (_BN_PAIR and _HIP_BN and _MFMA_LINEAR) else None
        if _BN_FOLD_CONV and _FUSE_ADD and n_valid is not None and ops.bn_finished(conv):
            # the convolution's statistics were finished by its contraction: its BatchNorm + LeakyReLU ride in unary2's
            # operand load (ops.bn_lrelu_linear) -- no normalising launch for the convolution at all
            if isinstance(us, UnaryBlock):
                sc = us(shortcut) if ys is None else us.batch_norm(ys, None)
            else:
                sc = shortcut
            out = self.unary2.forward_normalised(conv, n_valid, self.batch_norm_conv.batch_norm, 0.1, join=(sc, 0.1))
            if out is not None:
                return out
            shortcut = sc
            x = self.batch_norm_conv(conv, 0.1)
            return self.unary2(x, join=(shortcut, 0.1))
        if n_valid is not None and isinstance(us, UnaryBlock) and us.use_bn:
            # the BatchNorm of the convolution and the one of the shortcut are independent problems over the same rows:
            # one launch each way for the pair (ops.bn_lrelu_pair)
            if ys is None:
                ys = ops.linear(shortcut, us.mlp.weight, stats_n_valid=n_valid if _GEMM_STATS else None)
            x, shortcut = ops.bn_lrelu_pair(conv, self.batch_norm_conv.batch_norm, 0.1, ys, us.batch_norm.batch_norm, 1.0,
                                            n_valid)
        else:
            x = self.batch_norm_conv(conv, 0.1)
            shortcut = us(shortcut) if ys is None else us.batch_norm(ys, None)
        if _FUSE_ADD:       # x = unary2(x); return leaky_relu(x + shortcut)  (blocks.py:644-649), join fused
            return self.unary2(x, join=(shortcut, 0.1))
        return self.leaky_relu(self.unary2(x) + shortcut)


class GlobalAverageBlock(nn.Module):

    def forward(self, x, batch):
        return global_average(x, batch.lengths[-1])


class NearestUpsampleBlock(nn.Module):

    def __init__(self, layer_ind):
        super(NearestUpsampleBlock, self).__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return closest_pool(x, batch.upsamples[self.layer_ind - 1])

    def __repr__(self):
        return 'NearestUpsampleBlock(layer: {:d} -> {:d})'.format(self.layer_ind, self.layer_ind - 1)


class MaxPoolBlock(nn.Module):

    def __init__(self, layer_ind):
        super(MaxPoolBlock, self).__init__()
        self.layer_ind = layer_ind

    def forward(self, x, batch):
        return max_pool(x, batch.pools[self.layer_ind + 1])
