import json,sys
for f in sys.argv[1:]:
    j=json.loads([l for l in open(f) if l.startswith("{")][-1])
    w=j["roofline"]["work"]; rays=j["config"]["rays_per_frame"]["total"]
    print(f, j["value"], j["ms_per_step"], {k: round(w[k]/rays,2) for k in w if w[k]})
