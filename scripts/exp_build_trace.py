"""Builds the hg38-shaped index twice (for a rocprofv3 --kernel-trace --stats run: where bbidx_build's time goes)."""
import sys, time
sys.path.insert(0, ".")
import torch
import bench as B
from bbmap_amd.index import DeviceIndex
name = sys.argv[1] if len(sys.argv) > 1 else "hg38"
lens, paired, _ = B.WORKLOADS[name]
chroms, _ = B.shared_reference(name, lens, 0.0 if name == "ecoli" else 0.1, 0, 1)
for i in range(2):
    t = time.perf_counter()
    di = DeviceIndex.build(chroms, k=13)
    torch.cuda.synchronize()
    print("build %d: %.3f s" % (i, time.perf_counter() - t), flush=True)
    di.close()
