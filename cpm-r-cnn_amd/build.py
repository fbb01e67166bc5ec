"""Build libcpmrcnn_hip.so (gfx950) in-tree with hipcc.  No cmake, no torch: plain objects + link.

    python cpm-r-cnn_amd/build.py [--force] [--jobs N]
"""
import argparse
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib")
LIB = os.path.join(OUT, "libcpmrcnn_hip.so")

COMMON = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fvisibility=hidden", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc"]
# translation units whose arithmetic must round exactly like the reference's C++ (no FMA contraction)
EXACT = {"roi_align.hip", "nms.hip", "detect_glue.hip", "soft_nms.hip", "image_prep.hip", "roi_lists.hip"}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def newest_header():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(HERE, "..", "include", "cpmrcnn_hip.h"))
    return max(os.path.getmtime(h) for h in hs)


def compile_one(src, force):
    obj = os.path.join(OUT, "obj", src.replace(".hip", ".o"))
    path = os.path.join(CSRC, src)
    if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), newest_header()):
        return obj, False
    flags = list(COMMON)
    if src in EXACT:
        flags.append("-ffp-contract=off")
    cmd = ["hipcc"] + flags + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s\n%s" % (src, r.stdout, r.stderr))
    if r.stderr.strip():
        sys.stderr.write(r.stderr)
    return obj, True


def build(force=False, jobs=4):
    os.makedirs(os.path.join(OUT, "obj"), exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda s: compile_one(s, force), sources()))
    objs = [o for o, _ in res]
    if force or any(c for _, c in res) or not os.path.exists(LIB):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s\n%s" % (r.stdout, r.stderr))
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    print(build(a.force, a.jobs))
