import json,sys
for f in sys.argv[1:]:
    d=json.load(open(f))
    print(f, round(d["value"],2), "steps/s", round(d["ms_per_step"],2), "ms")
    print("   "+"  ".join(f"{k}:{v['avg_ms']*v['launches']/d['steps']:.2f}" for k,v in d["kernels"].items()))
