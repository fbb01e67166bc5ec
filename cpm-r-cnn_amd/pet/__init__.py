"""MI355X-native counterpart of the reference's `pet` package (hot path only).

Module paths, class names, registry keys, cfg keys and state-dict keys follow the reference so
that its YAML configs and entry points resolve against this tree; the arithmetic underneath is
libcpmrcnn_hip.so (hand-written HIP for gfx950), reached through `pet.lib.ops._hip`.
"""
