"""One-line digest of a bench.py JSON line read from stdin (value, ms/step, the GEMM roofline fields, the shares of attention / LayerNorm / FT sweep):
  python bench.py ... | python tools/debug/bench_pick.py <label>   -- used for the A/B runs of profiles/r03_summary.md U."""
import sys,json
j=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1])
r=j["roofline"]
print(sys.argv[1], "value", j["value"], "ms/step", j["ms_per_step"], "| gemm achieved", r["achieved"], "frac", r["frac"], "avg_us", r["avg_launch_us"], "launches", r["launches"],
      "gemm_time_frac", r["gemm_time_frac_of_step"], "exec TF/cycle", r["executed_tflop_per_cycle"], "steps", r.get("instrumented_steps"), "| att", j["kernels"]["attention"]["time_frac_of_step"],
      "ln", j["kernels"]["layernorm"]["time_frac_of_step"], "ft", j["roofline_hbm"]["ft_adamw_step"]["achieved"], j["roofline_hbm"]["ft_adamw_step"]["time_frac_of_step"])
