import json,sys
for line in sys.stdin:
    line=line.strip()
    if not line.startswith('{'): continue
    d=json.loads(line)
    keep={k:d.get(k) for k in ("value","ms_per_step")}
    keep["phases_ms"]=d.get("phases_ms")
    keep["roofline"]={k:d["roofline"].get(k) for k in ("achieved","frac","kernel_ms","alone_on_gpu_ms","alone_on_gpu") if k in d.get("roofline",{})}
    for k in ("sorted_search_mode","single_view","single_view_cfg1","host_buffer_path"):
        if k in d: keep[k]=d[k]
    print(json.dumps(keep))
