O=gpurun_out/gridsweep; mkdir -p $O
run() { # name shape envs...
  name=$1; sh=$2; shift 2
  timeout -k 10 200 env "$@" python bench.py --log-shape $sh --no-cpu --steps 20 --no-e2e --no-scatter-gather > $O/$name.json 2> $O/$name.err || { echo FAILED $name; exit 1; }
  python3 - $name $O/$name.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
k = list(d["kernel_ms"].values())
print(f"{sys.argv[1]:28s} step {d['ms_per_step']:7.3f} ms {d['value']:7.1f} GB/s k_anchor {k[0]:6.3f} tail {k[1]:6.3f} pipelined {d['pipelined']['value']:7.1f}", flush=True)
PY
}
run url_base url-heavy X=1
run url_v3 url-heavy MATCHY_AMD_GRID=0,3,0
run url_v5 url-heavy MATCHY_AMD_GRID=0,5,0
run url_lp128 url-heavy MATCHY_AMD_LPGRID=128
run url_lp512 url-heavy MATCHY_AMD_LPGRID=512
run url_lp1024 url-heavy MATCHY_AMD_LPGRID=1024
run jsonl_base jsonl-app X=1
run jsonl_lp512 jsonl-app MATCHY_AMD_LPGRID=512
run jsonl_tokaside0 jsonl-app MATCHY_AMD_TOK_ASIDE=0
run hash_base hash-dense X=1
run hash_tokaside1 hash-dense MATCHY_AMD_TOK_ASIDE=1
run hash_v4_1280 hash-dense MATCHY_AMD_MISC_GRID_V4=1280
run hash_v4_1792 hash-dense MATCHY_AMD_MISC_GRID_V4=1792
run nginx_base nginx X=1
