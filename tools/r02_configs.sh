#!/bin/bash
# usage (GPU box, repo root): tools/r02_configs.sh  -> gpurun_out/r02_*: batch sweep, the non-headline configs, encoder profile + counters
export TMPDIR=/tmp
bash tools/batch_sweep.sh > gpurun_out/r02_batch_sweep.jsonl 2> gpurun_out/r02_batch_sweep.err; echo "== sweep done"
python bench_configs.py --mode encode --batch 64 2> gpurun_out/r02_encode.err | tail -1 > gpurun_out/r02_cfg_encode.json; echo "== encode done"
python bench_configs.py --mode e2e 2> gpurun_out/r02_e2e.err | tail -1 > gpurun_out/r02_cfg_e2e.json; echo "== e2e done"
python bench_configs.py --mode cache 2> gpurun_out/r02_cache.err | tail -1 > gpurun_out/r02_cfg_cache.json; echo "== cache done"
python bench_configs.py --mode ingest 2> gpurun_out/r02_ingest.err | tail -1 > gpurun_out/r02_cfg_ingest.json; echo "== ingest done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r02_enc_prof -- python3 bench_configs.py --mode encode --batch 64 > /dev/null 2> gpurun_out/r02_enc_prof.err; echo "== encoder profile done"
bash tools/pmc_encode.sh r02enc > gpurun_out/r02_pmc_encode.txt 2>&1; echo "== encoder counters done"
python bench_configs.py --mode ivf 2> gpurun_out/r02_ivf.err | tail -1 > gpurun_out/r02_cfg_ivf.json; echo "== ivf done"
