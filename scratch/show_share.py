import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, round(d["share_s"], 2), "kernel", round(d["kernel_s"], 2), d["m_range"], {k: (round(v["seconds"], 2), round(v["kernel_s"], 2)) for k, v in d["stages"].items() if isinstance(v, dict)}, "other", round(d["stages"]["other_s"], 2))
    except Exception as e:
        print(f, "ERR", repr(e))
