set -e
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gpu_tests.log 2>&1 || { tail -30 gpurun_out/gpu_tests.log; exit 1; }
tail -2 gpurun_out/gpu_tests.log
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d["roofline"]["by_kernel"]; print(d["ms_per_step"], k["k_extend<true>"]["ms"], k["k_extend<false>"]["ms"], k["k_shadow"]["ms"], k["k_shade"]["ms"])'
for w in config3 config3u config3 config3u; do echo $w; timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --workload $w | python -c "$P"; done
