"""utils.metrics of the reference under its own names (KPConv-PyTorch/utils/metrics.py:35-80 fast_confusion,
:206-232 IoU_from_confusions), NumPy in / NumPy out like the reference; tensors resident in HBM are
accepted too and then counted on the GPU (utils/voting.py holds the device versions)."""
import numpy as np
import torch

try:
    from . import voting
except ImportError:
    import voting


def _as_labels(a, what):
    if isinstance(a, torch.Tensor):
        a = a.squeeze()
        if a.dim() != 1:
            raise ValueError('{:s} values are stored in a {:d}D array instead of 1D array'.format(what, a.dim()))
        if a.dtype not in (torch.int32, torch.int64):
            raise ValueError('{:s} values are {:s} instead of int32 or int64'.format(what, str(a.dtype)))
        return a
    a = np.squeeze(a)
    if len(a.shape) != 1:
        raise ValueError('{:s} values are stored in a {:d}D array instead of 1D array'.format(what, len(a.shape)))
    if a.dtype not in [np.int32, np.int64]:
        raise ValueError('{:s} values are {:s} instead of int32 or int64'.format(what, str(a.dtype)))
    return a


def fast_confusion(true, pred, label_values=None):
    """Confusion matrix, rows = truth, columns = prediction, classes = sorted label_values (default: the
    labels that occur). Arbitrary non-negative label values are compacted through a lookup table."""
    true, pred = _as_labels(true, 'Truth'), _as_labels(pred, 'Prediction')
    on_device = isinstance(true, torch.Tensor) and true.is_cuda
    if label_values is None:
        both = torch.cat([torch.as_tensor(true).reshape(-1), torch.as_tensor(pred).reshape(-1)]).cpu().numpy()
        label_values = np.unique(both)
    else:
        label_values = np.asarray(label_values)
        if label_values.dtype not in [np.int32, np.int64]:
            raise ValueError('label values are {:s} instead of int32 or int64'.format(str(label_values.dtype)))
        if len(np.unique(label_values)) < len(label_values):
            raise ValueError('Given labels are not unique')
    label_values = np.sort(label_values)
    num_classes = len(label_values)
    if label_values[0] < 0:
        raise ValueError('Unsupported negative classes')
    dev = true.device if on_device else torch.device('cpu')
    t, p = torch.as_tensor(true).to(dev).long(), torch.as_tensor(pred).to(dev).long()
    if not (label_values[0] == 0 and label_values[-1] == num_classes - 1):
        table = np.zeros((int(label_values[-1]) + 1,), dtype=np.int64)
        table[label_values] = np.arange(num_classes)
        table = torch.from_numpy(table).to(dev)
        t, p = table[t], table[p]
    conf = voting.confusion(t, p, num_classes)
    return conf if on_device else conf.numpy()


def IoU_from_confusions(confusions):
    """[..., C, C] -> [..., C]; classes absent from the truth get the mean IoU of the present ones."""
    if isinstance(confusions, torch.Tensor):
        return voting.IoU_from_confusions(confusions)
    return voting.IoU_from_confusions(np.asarray(confusions)).numpy()
