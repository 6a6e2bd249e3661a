import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); k = d["render_loss_step_ms"]["kernels"]
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"]), round(d["ms_per_step"], 4), "| train", d["train_step_with_head"].get("ms_per_step"), "| e2e", d.get("train_step_e2e_standin", {}).get("ms_per_step", d.get("train_step_e2e_standin")), "|", {n: round(v["avg_ms"], 4) for n, v in k.items()})
