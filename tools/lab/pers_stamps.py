"""Where a tile pass of the persistent update spends its time: shader-clock stamps of wave 0 at the first barrier of every
K stage (-DPERS_STAMPS build: bash tools/lab/exp_variants.sh build20), first 8 tiles of every workgroup, one launch.
   CIMRGP_LIB_PATH=cimrgp_amd/libcimrgp_tuning_e20.so python tools/lab/pers_stamps.py [m]"""
import ctypes, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from cimrgp_amd import device as dev, _lib

m = int(sys.argv[1]) if len(sys.argv) > 1 else 7936
k = 256
dev.require_gpu()
lib = _lib.load()
c = dev.alloc_matrix(m, m, torch.float64, "cuda"); a = dev.alloc_matrix(m, k, torch.float64, "cuda")
c.normal_(); a.normal_()
for _ in range(3):
    dev.syrk_lower(c, a, m, k)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); dev.syrk_lower(c, a, m, k); e1.record(); torch.cuda.synchronize()
buf = (ctypes.c_longlong * (256 * 132))()
lib.cimrgp_debug_pers_stamps.restype = ctypes.c_int
assert lib.cimrgp_debug_pers_stamps(buf) == 0
raw = np.frombuffer(buf, dtype=np.int64).reshape(256, 132)
full = raw[:, 127] != 0                              # workgroups that ran at least 8 tiles
entry_to_first = raw[full, 0] - raw[full, 128]       # kernel entry -> first stage's barrier (C tile + first operands)
lifetime = raw[full, 129] - raw[full, 128]
clock_ghz = np.median(lifetime / np.maximum(1, raw[full, 131] - raw[full, 130])) * 0.1      # s_memrealtime ticks at 100 MHz
st = raw[full, :128].reshape(-1, 8, 16)
d = np.diff(st.reshape(-1, 128), axis=1)            # clocks between consecutive stage barriers
d = d[:, :127]
ideal = 2 * 32 * 64                                  # two waves per SIMD, 32 multiplies of 64 clocks per stage and wave
print(json.dumps(dict(m=m, launch_us=round(e0.elapsed_time(e1) * 1e3, 1), ideal_clocks_per_stage=ideal,
                      median=float(np.median(d)), mean=float(d.mean()), p10=float(np.percentile(d, 10)), p90=float(np.percentile(d, 90)),
                      p99=float(np.percentile(d, 99)), max=float(d.max()),
                      frac_over_1p25x=float((d > 1.25 * ideal).mean()), frac_over_2x=float((d > 2 * ideal).mean()),
                      per_stage_position_median=[float(v) for v in np.median(np.diff(st, axis=2).reshape(-1, 15), axis=0)],
                      tile_boundary_median=float(np.median(st[:, 1:, 0] - st[:, :-1, 15])),
                      workgroups_with_8_tiles=int(full.sum()), entry_to_first_barrier_median=float(np.median(entry_to_first)), entry_to_first_barrier_max=float(entry_to_first.max()), lifetime_median=float(np.median(lifetime)), in_kernel_clock_ghz=float(clock_ghz), lifetime_max=float(lifetime.max()),
                      pass_clocks_median=float(np.median(st[:, 1:, 0] - st[:, :-1, 0])))))
