mkdir -p gpurun_out/rp2
for c in 1 2 4; do for f in d 0 1 2; do
  if [ $f = d ]; then unset MSQ_GEMM_256; else export MSQ_GEMM_256=$f; fi
  timeout 300 python3 bench.py --single-rank-collectives --steps 10 --warmup 2 --chunks $c --no-cpu-baseline > gpurun_out/rp2/c${c}_f${f}.json 2> gpurun_out/rp2/c${c}_f${f}.err
done; done
unset MSQ_GEMM_256
python3 - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/rp2/*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][0]); r=d["rowparallel"]
        print(f.split("/")[-1], "step %.3f gemm %.3f comm %.3f exposed %.3f  tflops %.0f" % (r["step_ms"], r["gemm_ms"], r["comm_ms"] or 0, r["exposed_comm_ms"], r["tflops_whole_job"]))
    except Exception as e: print(f, "ERR", e)
PY
