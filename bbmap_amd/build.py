"""Builds libbbmap_amd.so (HIP, gfx950 only) in-tree with hipcc: one object per source file, stale ones only, in parallel."""
import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
OUT = os.path.join(HERE, "libbbmap_amd.so")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "bbmap_amd.h"))
    return hs


def _flags():
    return ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
            "-I" + os.path.join(ROOT, "include"), "-I" + CSRC] + os.environ.get("BBMSA_CXXFLAGS", "").split()


def _obj_path(src, flags):
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:8]          # another flag set gets its own objects
    return os.path.join(OBJ, os.path.basename(src)[:-4] + "." + tag + ".o")


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(s) > t for s in sources() + headers())


def build(force=False, verbose=False):
    flags = _flags()
    stamp = os.path.join(HERE, ".build_flags")          # travels with the .so (the objects do not)
    same_flags = os.path.exists(stamp) and open(stamp).read() == " ".join(flags)
    if not force and same_flags and not is_stale():
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    hdr_t = max(os.path.getmtime(h) for h in headers())
    todo, objs = [], []
    for s in sources():
        o = _obj_path(s, flags)
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
            todo.append((s, o))

    def compile_one(so):
        cmd = [hipcc] + flags + ["-c", so[0], "-o", so[1]]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    workers = max(1, min(len(todo), int(os.environ.get("BBMSA_BUILD_JOBS", "6"))))
    if todo:
        with cf.ThreadPoolExecutor(workers) as ex:
            list(ex.map(compile_one, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", OUT]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    with open(stamp, "w") as f:
        f.write(" ".join(flags))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
