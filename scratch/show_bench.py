import json,sys
d=json.loads(open(sys.argv[1]).read().strip())
print(d["value"], d["ms_per_step"], d["stage_ms"], d["parity"]["green"], d["parity"]["ev_max_err_over_pencil_tol"], d["parity"]["sv_max_err_over_svmax"], d["parity"]["svnum_equal"], d["parity"]["kept_counts_equal"])
print({k: round(v["ms_per_step"],2) for k,v in d["roofline"]["classes"].items()})
