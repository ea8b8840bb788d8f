"""Builds bbmap_amd/_variants/lib_<name>.so for an experiment: the listed source files compiled with extra flags, every other object
taken from the default build (bbmap_amd/csrc/_obj).  Load it with BBMAP_AMD_SO=<path> (bbmap_amd/_lib.py).
python scripts/build_variant.py occ6 "-DBBIDX_LONG_SHORT_OCC=6 -DBBIDX_CYC_MAPW=128" index_probe_wave.hip"""
import os, subprocess, sys
sys.path.insert(0, ".")
from bbmap_amd import build as B

name, extra, files = sys.argv[1], sys.argv[2].split(), sys.argv[3:]
B.build()
flags = B._flags()
out_dir = os.path.join(B.HERE, "_variants")
os.makedirs(out_dir, exist_ok=True)
objs = []
for s in B.sources():
    if os.path.basename(s) in files:
        o = os.path.join(out_dir, "%s.%s.o" % (os.path.basename(s)[:-4], name))
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + extra + ["-c", s, "-o", o])
        objs.append(o)
    else:
        objs.append(B._obj_path(s, flags))
out = os.path.join(out_dir, "lib_%s.so" % name)
subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
print(out)
