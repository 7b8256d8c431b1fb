#!/bin/bash
# round-5 GPU call 3: where a stage of the persistent update spends its clocks (stamped variants), in-kernel clock
mkdir -p gpurun_out
bash tools/lab/exp_variants.sh run20 7936 2>&1 | tee gpurun_out/r05_pers_stamps_variants.txt
bash tools/lab/exp_variants.sh run20 6912 2>&1 | tee -a gpurun_out/r05_pers_stamps_variants.txt
