#!/bin/bash
# timing-only variants of the persistent kernel (libcimrgp_tuning_eN.so: wrong results by construction)
cd "$(dirname "$0")/../.."
# build the variants first (on the CPU box, before gpurun):  bash tools/lab/exp_variants.sh build
if [ "$1" = build20 ]; then       # stamped builds (tools/lab/pers_stamps.py): s0 = the shipped kernel, s1 = no second barrier, s2 = no barriers, s3 = no C events (1-3: wrong results)
  bash tools/build_tuning.sh
  cd cimrgp_amd/csrc
  for e in 0 1 2 3; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCIMRGP_TUNING -DPERS_STAMPS -DPERS_EXP=$e -c gemm_nt.hip -o tuning_obj/gemm_nt_s$e.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcimrgp_tuning_s$e.so tuning_obj/api.o tuning_obj/gemm_nt_s$e.o \
      tuning_obj/potrf.o tuning_obj/gram.o tuning_obj/solve.o tuning_obj/misc.o tuning_obj/reduced.o tuning_obj/layer.o tuning_obj/comm.o -ldl
  done
  exit 0
fi
if [ "$1" = run20 ]; then
  for e in 0 1 2 3; do
    echo "== stamped variant s$e"
    CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning_s$e.so CIMRGP_GEMM_PERS=256 timeout -k 10 100 python3 tools/lab/pers_stamps.py ${2:-7936} 2>&1 | grep -v amdgpu.ids
  done
  exit 0
fi
if [ "$1" = build ]; then
  bash tools/build_tuning.sh
  cd cimrgp_amd/csrc
  for e in 1 2 3 4 5 6 7 8 9 10 11; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -DCIMRGP_TUNING -DPERS_EXP=$e \
      -c gemm_nt.hip -o tuning_obj/gemm_nt_e$e.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libcimrgp_tuning_e$e.so tuning_obj/api.o tuning_obj/gemm_nt_e$e.o \
      tuning_obj/potrf.o tuning_obj/gram.o tuning_obj/solve.o tuning_obj/misc.o tuning_obj/reduced.o tuning_obj/layer.o tuning_obj/comm.o -ldl
  done
  exit 0
fi
for v in "" _e1 _e2 _e3 _e4 _e5 _e6 _e7 _e8; do
  echo "== variant ${v:-base}"
  CIMRGP_LIB_PATH=$PWD/cimrgp_amd/libcimrgp_tuning$v.so CIMRGP_GEMM_PERS=256 timeout -k 10 100 python3 tools/gemm_bench.py --m 7936,5888 --k 256 --reps 20 --check 2>&1 | grep -v amdgpu.ids
done
