"""Print the essentials of a bench.py JSON line: value, ms/step and the top kernels.  usage: bench_brief.py file [tag]"""
import json
import sys

path = sys.argv[1]
tag = sys.argv[2] if len(sys.argv) > 2 else path
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    top = list(d["roofline"]["kernels_ms_per_step"].items())[:10]
    print(tag, d["value"], "img/s", d["ms_per_step"], "ms/step", "frac", d["roofline"]["frac"], dict(top))
    if "parity" in d:
        print("   parity", d["parity"])
    if "modes" in d:
        print("   modes", d["modes"])
    for name, st in d.get("stages", {}).items():
        print("   stage", name, {k: v for k, v in st.items() if k not in ("workload", "kernels_ms")})
    if "cpu_baseline" in d:
        print("   cpu_baseline", d["cpu_baseline"])
except Exception as exc:          # noqa: BLE001
    print(tag, "no bench line:", exc)
    try:
        print(open(path.replace(".json", ".err")).read()[-1500:])
    except OSError:
        pass
