"""Compile the reference's own CPU RoIAlign and soft-NMS into oracle/_ref/ (git-ignored).

Recipe only: sources are read from /root/reference where they lie, nothing is
copied.  Runs in the build container; the GPU box ships the prebuilt .so.
TEST INFRASTRUCTURE ONLY.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/pet/lib/ops/csrc"


def build(verbose=False):
    out = os.path.join(HERE, "_ref")
    if not os.path.isdir(REF):
        return None
    os.makedirs(out, exist_ok=True)
    os.environ.setdefault("MAX_JOBS", "4")
    from torch.utils.cpp_extension import load
    # a CPU-only C++ build: keep hipcc out of it
    return load(name="cpm_ref", sources=[os.path.join(REF, "ROIAlign", "ROIAlign_cpu.cpp"),
                                         os.path.join(REF, "NMS", "soft_nms.cpp"),
                                         os.path.join(REF, "NMS", "ml_soft_nms.cpp"),
                                         os.path.join(HERE, "ref_binding.cpp")],
                extra_include_paths=[os.path.join(REF, "ROIAlign"), os.path.join(REF, "NMS")],
                extra_cflags=["-O2", "-ffp-contract=off"],
                build_directory=out, verbose=verbose, with_cuda=False)


def load_prebuilt():
    """Import oracle/_ref/cpm_ref.so if it exists (GPU box), else None."""
    import importlib.util
    import torch  # noqa: F401  (libtorch must be loaded first)
    so = os.path.join(HERE, "_ref", "cpm_ref.so")
    if not os.path.exists(so):
        return None
    spec = importlib.util.spec_from_file_location("cpm_ref", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    m = build(verbose="-v" in sys.argv)
    print("built" if m is not None else "reference not present; nothing built")
