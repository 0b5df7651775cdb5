import json,sys
d=json.load(open(sys.argv[1]))
c=d["config"]
print(sys.argv[1], d["value"], d["ms_per_step"])
for k in ("newton_iterations","newton_linear_applications","substep_s"):
    print(" ", k, c[k])
