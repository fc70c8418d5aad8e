"""Fall-through for the mirrored packages: names the drop-in does not provide resolve to the
reference's own modules.

``dropin/`` mirrors only the hot-path modules of the reference's packages (``datasets``, ``utils``,
``models``, ``kernels``, ``cpp_wrappers``, ``mvpnet``, ``common``). Once it precedes the reference on
``sys.path`` those package names belong to the drop-in, so every mirrored ``__init__`` appends the
same-named directories found LATER on ``sys.path`` (and in the current directory, which stands for
``sys.path[0] == ''`` of a script run from the reference tree) to its ``__path__``: the drop-in's
modules win, everything else (``datasets.ScanNet_sphere_color``, ``utils.trainer``,
``utils.mayavi_visu``, ``mvpnet.utils.visualize``, ``common.utils.*`` ...) is imported from the
reference unchanged. All of the reference's own ``__init__.py`` files of these packages are empty,
so nothing is lost by not executing them.
"""
import os
import sys

_DROPIN = os.path.dirname(os.path.abspath(__file__))


def extend(pkg_path, pkg_name, pkg_file):
    """pkg_path: the package's ``__path__`` (extended in place); only when the package was imported as a
    top-level mirror (``datasets``), not as ``<package>.dropin.datasets``."""
    here = os.path.dirname(os.path.abspath(pkg_file))
    rel = os.path.relpath(here, _DROPIN)
    if pkg_name != rel.replace(os.sep, "."):
        return pkg_path
    seen = {os.path.realpath(p) for p in pkg_path}
    for entry in list(sys.path):
        base = os.path.abspath(entry or os.getcwd())
        if os.path.realpath(base) == os.path.realpath(_DROPIN):
            continue
        cand = os.path.join(base, rel)
        if os.path.isdir(cand) and os.path.realpath(cand) not in seen:
            seen.add(os.path.realpath(cand))
            pkg_path.append(cand)
    return pkg_path


def reference_module_file(rel_py):
    """Path of the reference's own ``rel_py`` (e.g. 'utils/config.py') further along ``sys.path``, or None."""
    for entry in list(sys.path):
        base = os.path.abspath(entry or os.getcwd())
        if os.path.realpath(base) == os.path.realpath(_DROPIN):
            continue
        cand = os.path.join(base, rel_py)
        if os.path.isfile(cand):
            return cand
    return None


def worker_start_method():
    """Called by the mirrored ``datasets`` package (route 1 only). The reference builds its batches -- and with them
    the input pyramid -- in ``config.input_threads`` DataLoader workers that the script creates with the platform's
    default start method (train_ScanNet_sphere.py:365-377: no ``multiprocessing_context``), i.e. ``fork`` on Linux,
    after the model has initialised the GPU in the parent. A HIP context does not survive ``fork`` (the library
    refuses such a call, ``_lib.FORK_MESSAGE``), so the drop-in makes ``spawn`` the default start method before the
    script creates its loaders: every worker is then a fresh interpreter with its own HIP context, the script stays
    unmodified. Only a start method nobody has chosen yet is set: one the host script picked explicitly
    (``multiprocessing.set_start_method`` before importing ``datasets``) is left alone, with a warning when it is
    ``fork``. ``MVK_DATALOADER_START=keep`` never touches the start method; ``fork`` / ``forkserver`` / ``spawn`` in that
    variable select one explicitly (and then override an earlier choice: the variable is the operator's word)."""
    import multiprocessing as mp
    import warnings
    want = os.environ.get("MVK_DATALOADER_START")
    current = mp.get_start_method(allow_none=True)
    if want == "keep":
        return current
    if want is None:
        if current is None:
            mp.set_start_method("spawn")
            return "spawn"
        if current == "fork":
            warnings.warn("the multiprocessing start method was set to 'fork' before the MV-KPConv drop-in was imported: "
                          "DataLoader workers that call into the HIP library will fail (a HIP context does not survive "
                          "fork); use 'spawn' or 'forkserver', or pass multiprocessing_context to the loaders")
        return current
    if current != want:
        mp.set_start_method(want, force=True)
    return want
