cd $GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/r04d_bench_c4.json 2> gpurun_out/r04d_bench_c4.err
(timeout 1200 python -m pytest tests/test_bench_multi_gpu.py tests/test_abi_cpu.py -x -q 2>&1 | tail -15) > gpurun_out/r04d_bench_tests.log
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04d_bench_c4.json"))
print(d["value"], d["ms_per_step"], {k:d["roofline"][k] for k in ("bound","achieved","peak","frac","traffic")}, d["roofline"].get("whole_step_traffic"), d["roofline"]["mfma"]["frac"])
print(json.dumps(d.get("lfplus"))[:1500])
print(json.dumps(d.get("cpu_baseline"))[:400])
PY
tail -5 gpurun_out/r04d_bench_c4.err; cat gpurun_out/r04d_bench_tests.log
