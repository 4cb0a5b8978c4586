#!/bin/bash
# Collects the per-round measurement set on the GPU box (run through gpurun from the repo root):
#   tools/collect_round_profiles.sh r02a
# -> gpurun_out/<tag>_*: default bench line, bench under rocprofv3 --stats (+ kernel stats CSV), HBM traffic
#    (FETCH_SIZE / WRITE_SIZE passes), SQ/LDS counters per kernel, per-phase counters and phase ablation of the
#    fast fingerprint kernel.  Copy what should be judged into profiles/.
TAG=${1:-r05h}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
python3 bench.py > gpurun_out/${TAG}_bench.log 2> gpurun_out/${TAG}_bench.err
D=gpurun_out/${TAG}_stats; rm -rf $D
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$D -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu --no-secondary) > gpurun_out/${TAG}_bench_under_rocprof.log 2>&1
cp $(find $D -name '*kernel_stats.csv' | head -1) gpurun_out/${TAG}_kernel_stats.csv 2>/dev/null
python3 tools/collect_traffic.py --tag ${TAG} --out gpurun_out/${TAG}_traffic > gpurun_out/${TAG}_traffic.log 2>&1
cp profiles/traffic.json gpurun_out/${TAG}_traffic.json 2>/dev/null
python3 tools/collect_sq.py --tag ${TAG} --out gpurun_out/${TAG}_sq > gpurun_out/${TAG}_sq.log 2>&1
python3 tools/phase_counters.py 262144 > gpurun_out/${TAG}_phase_counters.txt 2> gpurun_out/${TAG}_phase_counters.err
python3 tools/profile_fingerprint.py 16384 1 1000000 > gpurun_out/${TAG}_fast_kernel_phase_shares.txt 2>&1
python3 tools/long_window_bench.py 110 15 30 8192 > gpurun_out/${TAG}_window_lengths.txt 2>&1
python3 tools/bench_refine.py 32768 > gpurun_out/${TAG}_refine.txt 2>&1
bash tools/dev/rfstats.sh > gpurun_out/${TAG}_refine_kernel_trace.txt 2>&1
python3 tools/pmc_kernel.py refine_tail_kernel,refine_match_wave,fingerprint_list_kernel -- python3 $PWD/tools/bench_refine.py 32768 > gpurun_out/${TAG}_refine_sq_counters.txt 2>&1
python3 bench.py --leg shipped_model_e2e > gpurun_out/${TAG}_shipped_model_e2e.json 2> /dev/null
for t in "110 6 12" "110 15 30" "120 9 18"; do python3 tools/profile_exact.py $t 65536 >> gpurun_out/${TAG}_exact_kernel_triples.txt 2>&1; done
for t in "110 15 30 65536 2.5" "110 15 30 262144 1.0" "120 9 18 262144 1.0"; do
    n=$(echo $t | tr ' .' '__'); tools/dev/kstats.sh ${TAG}_triple_$n python3 $PWD/tools/bench_triple.py $t 5 >> gpurun_out/${TAG}_triples_kernel_split.txt 2>&1
done
for w in 4 16; do python3 tools/host_workers.py --workers $w --mode feeder --seconds 2 >> gpurun_out/${TAG}_host_workers.txt 2>&1; python3 tools/host_workers.py --workers $w --mode sync --seconds 2 >> gpurun_out/${TAG}_host_workers.txt 2>&1; done
# the C4 launch path on this one-GPU box (eight ranks share the device, host collectives): a same-tree reference for the
# first real 8-GPU run's per_rank_ms_per_step / rccl_ranks
WDX_BENCH_BACKEND=gloo timeout 600 python3 bench.py --gpus 8 --reads 20000 --steps 2 --warmup 1 --no-cpu --no-secondary 2> gpurun_out/${TAG}_c4_gloo.err | grep '^{' > gpurun_out/${TAG}_c4_gloo_8ranks_one_gpu.json
rm -f gpurun_out/${TAG}_triple_*_kstats.log gpurun_out/${TAG}_triple_*_kernel_stats.csv
rm -rf gpurun_out/${TAG}_stats gpurun_out/${TAG}_traffic gpurun_out/${TAG}_sq gpurun_out/phase_pmc
ls -la gpurun_out | grep ${TAG}
