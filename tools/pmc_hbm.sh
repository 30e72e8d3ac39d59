#!/bin/bash
# HBM traffic of the dominant kernels from the PMC counters (two separate passes, kernel-trace only) ->
# profiles/<TAG>_pmc_hbm_raw.txt.  usage: tools/pmc_hbm.sh TAG
TAG=${1:-r01}
for c in FETCH_SIZE WRITE_SIZE; do
  /usr/local/graft/bin/gpurun --timeout 900 -- "mkdir -p gpurun_out/${TAG}_$c; export TMPDIR=/tmp; rocprofv3 --pmc $c --kernel-trace -d gpurun_out/${TAG}_$c -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/${TAG}_$c/bench.log 2>&1" 2>&1 | tail -1
done
{
  echo "# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 2 --no-cpu-baseline"
  echo "# mean per dispatch; counter unit as reported (KB)"
  for c in FETCH_SIZE WRITE_SIZE; do
    echo "## $c"
    python tools/pmc_agg.py gpurun_out/${TAG}_$c/p_results.db | grep -E "n dur_us|recon_nll|small_up|igemm_bf_kernel<128, 32, 3|smallcin" | sed 's/M  / /g'
  done
} > profiles/${TAG}_pmc_hbm_raw.txt
cat profiles/${TAG}_pmc_hbm_raw.txt
