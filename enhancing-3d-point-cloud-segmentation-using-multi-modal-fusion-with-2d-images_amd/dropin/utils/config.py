"""Plain attribute-bag configuration with the reference's field names and defaults
(KPConv-PyTorch/utils/config.py:35-279). Only the fields the hot path reads are kept; the
txt (de)serialiser, dataset paths and visualisation switches are out of scope."""
import numpy as np


class bcolors:
    """ANSI escape codes the reference's console messages use (utils/config.py:24-32)."""
    HEADER, OKBLUE, OKGREEN, WARNING, FAIL = '\033[95m', '\033[94m', '\033[92m', '\033[93m', '\033[91m'
    ENDC, BOLD, UNDERLINE = '\033[0m', '\033[1m', '\033[4m'


class Config:
    # input / task
    dataset = ''
    dataset_task = ''
    num_classes = 0
    in_points_dim = 3
    in_features_dim = 1
    in_radius = 1.0
    input_threads = 8

    # architecture
    architecture = []
    equivar_mode = ''
    invar_mode = ''
    first_features_dim = 64
    use_batch_norm = True
    batch_norm_momentum = 0.99
    segmentation_ratio = 1.0

    # KPConv
    num_kernel_points = 15
    first_subsampling_dl = 0.02
    conv_radius = 2.5
    deform_radius = 5.0
    KP_extent = 1.0
    KP_influence = 'linear'
    aggregation_mode = 'sum'
    fixed_kernel_points = 'center'
    modulated = False
    n_frames = 1
    max_in_points = 0
    max_val_points = 50000
    val_radius = 51.0

    # fusion variant switches (config.py:91-93)
    early_fusion = False
    middle_fusion = False
    late_fusion = False
    path_2D = ''

    # training
    learning_rate = 1e-3
    momentum = 0.9
    lr_decays = {200: 0.2, 300: 0.2}
    grad_clip_norm = 100.0
    augment_scale_anisotropic = True
    augment_symmetries = [False, False, False]
    augment_rotation = 'vertical'
    augment_scale_min = 0.9
    augment_scale_max = 1.1
    augment_noise = 0.005
    augment_color = 0.7
    augment_occlusion = 'none'
    weight_decay = 1e-3
    segloss_balance = 'none'
    class_w = []
    deform_fitting_mode = 'point2point'
    deform_fitting_power = 1.0
    deform_lr_factor = 0.1
    repulse_extent = 1.0
    batch_num = 10
    val_batch_num = 10
    max_epoch = 1000
    epoch_steps = 1000
    validation_size = 100
    checkpoint_gap = 50
    saving = True
    saving_path = None

    def __init__(self):
        # number of layers / which layers are deformable (config.py:237-279)
        self.num_layers = len([b for b in self.architecture if 'pool' in b or 'strided' in b]) + 1
        layer_blocks, self.deform_layers = [], []
        for block in self.architecture:
            if not ('pool' in block or 'strided' in block or 'global' in block or 'upsample' in block):
                layer_blocks.append(block)
                continue
            deform = bool(layer_blocks) and bool(np.any(['deformable' in b for b in layer_blocks]))
            if 'pool' in block or 'strided' in block:
                deform = deform or 'deformable' in block
            self.deform_layers.append(deform)
            layer_blocks = []
            if 'global' in block or 'upsample' in block:
                break


def _adopt_reference_config():
    """When dropin/ shadows a reference tree (INTEGRATION.md route 1), ``utils.config`` must stay the
    reference's own module -- dataset paths, save() / load() of parameters.txt, every field of its train
    scripts -- because nothing in it is on the hot path. The class above is only the stand-alone bag the
    synthetic harness uses when no reference is present."""
    if __name__ != "utils.config":
        return
    try:
        from _fallthrough import reference_module_file
    except ImportError:
        return
    path = reference_module_file("utils/config.py")
    if path is None:
        return
    import importlib.util
    spec = importlib.util.spec_from_file_location("utils._reference_config", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for k, v in vars(mod).items():
        if not k.startswith("__"):
            globals()[k] = v


_adopt_reference_config()
