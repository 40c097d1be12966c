"""Developer probe: isolate a hang of the precision-32 provider with transpose-view operands (RMHIP_TRACE=1)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = {
    "unary_view_64": "h=p.transpose(p.upload(np.ones((64,64)).reshape(-1),(64,64))); o=p.unary_sin(h); p.synchronize(); print(p.download(o)[:2])",
    "add_view_64": "h=p.transpose(p.upload(np.ones((64,64)).reshape(-1),(64,64))); o=p.elem_add(h,h); p.synchronize(); print(p.download(o)[:2])",
    "fused_view_64": "h=p.transpose(p.upload(np.ones((64,64)).reshape(-1),(64,64))); o=p.fused_elementwise(sh,[h,h,h],(64,64),4096); p.synchronize(); print(p.download(o)[:2])",
    "fused_plain_512": "h=p.upload(np.ones((512,512)).reshape(-1),(512,512)); o=p.fused_elementwise(sh,[h,h,h],(512,512),262144); p.synchronize(); print(p.download(o)[:2])",
    "fused_view_512": "h=p.transpose(p.upload(np.ones((512,512)).reshape(-1),(512,512))); o=p.fused_elementwise(sh,[h,h,h],(512,512),262144); p.synchronize(); print(p.download(o)[:2])",
    "fused_pyupload_512": "hs=[p.upload(np.ones((512,512))) for _ in range(3)]; o=p.fused_elementwise(sh,hs,(512,512),262144); p.synchronize(); print(p.download(o)[:2])",
}
PRE = ("import sys, numpy as np; sys.path.insert(0, %r); from runmat_amd import HipProvider; "
       "from runmat_amd.fusion import sin_mul_add_plan; plan,out=sin_mul_add_plan(); sh=plan.generate_wgsl_for_output(out,'f32'); "
       "p=HipProvider(0, precision='F32'); " % ROOT)
for name, body in CASES.items():
    env = dict(os.environ, RMHIP_TRACE="1")
    try:
        r = subprocess.run([sys.executable, "-c", PRE + body], env=env, capture_output=True, text=True, timeout=25)
        print(f"== {name}: rc {r.returncode}\n{r.stdout[-300:]}{r.stderr[-1500:]}", flush=True)
    except subprocess.TimeoutExpired as e:
        err = e.stderr.decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
        print(f"== {name}: TIMEOUT\n{err[-1500:]}", flush=True)
