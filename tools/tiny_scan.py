import sys, ctypes as C
sys.path.insert(0,'.')
import matchy_amd as M
from tools import synth
cfg=synth.config("c2")
db=M.Database(synth.build_db(cfg))
sc=M.Scanner(db, profile=True)
hip=C.CDLL("libamdhip64.so")
for n_lines in (10, 100000, 1000000):
    text=synth.make_log(cfg,0,n_lines)
    d=C.c_void_p(); hip.hipMalloc(C.byref(d), len(text)+64); hip.hipMemcpy(d, text, len(text), 1)
    for it in range(6):
        r=sc.scan_device(d.value, len(text), stream=0, fetch_mode=1); r.close()
    print(n_lines, len(text), {k:round(v,4) for k,v in sc.timing_ms().items()})
