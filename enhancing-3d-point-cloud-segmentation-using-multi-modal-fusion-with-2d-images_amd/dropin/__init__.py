"""Host-side mirror of the reference's module layout for the hot path.

Put this directory first on ``sys.path`` and the reference's own entry scripts resolve
``cpp_wrappers.*``, ``kernels.kernel_points``, ``models.blocks``, ``models.architectures*``,
``mvpnet.ops.group_points``, ``mvpnet.models.mvpnet_3d`` and ``common.nn`` to the MI355X-native
implementations. Every module imports the package's ``ops`` through
``_native`` so that it works both as ``<package>.dropin.<module>`` and as a top-level module.
"""
