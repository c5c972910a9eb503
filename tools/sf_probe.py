import sys, ctypes as C, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from atdn_vslam_amd import _lib
L = _lib.lib()
vp = lambda t: C.c_void_p(t.data_ptr())
cases = [(64,64,3,3,1,1,1,23,37,2),(128,256,1,5,1,0,2,47,154,2),(512,256,1,5,1,0,2,47,154,1),(256,192,3,3,1,1,1,47,154,1),(128,126,3,3,1,1,1,20,64,1),(96,128,1,1,2,0,0,31,45,2),(352,256,1,1,1,0,0,47,154,1)]
for (cin,cout,kh,kw,st,ph,pw,H,W,nimg) in cases:
    r = np.random.RandomState(1)
    x = torch.from_numpy(r.normal(0,1,(nimg,cin,H,W)).astype(np.float32))
    w = torch.from_numpy((r.uniform(-1,1,(cout,cin,kh,kw))*np.sqrt(3.0/(cin*kh*kw))).astype(np.float32))
    b = torch.from_numpy(r.uniform(-.5,.5,(cout,)).astype(np.float32))
    ref64 = F.conv2d(x.double(), w.double(), b.double(), stride=st, padding=(ph,pw))
    ref32 = F.conv2d(x, w, b, stride=st, padding=(ph,pw))
    xd = x.permute(0,2,3,1).contiguous().cuda()
    ho, wo = ref32.shape[2:]
    outs = {}
    for name, fn, extra in (("f32", L.atdn_conv2d_nhwc, (0,)), ("sf", L.atdn_conv2d_nhwc_sf, ())):
        o = torch.full((nimg,ho,wo,cout), float("nan"), device="cuda")
        args = [vp(xd), nimg, H, W, cin, vp(w), vp(b), cout, kh, kw, st, ph, pw] + list(extra) + [vp(o), None]
        _lib.check(fn(*args)); torch.cuda.synchronize()
        outs[name] = o.cpu().permute(0,3,1,2).double()
    e = lambda a: float((a-ref64).abs().max())
    print(cin,cout,kh,kw, "cpu32 err %.2e  f32mfma err %.2e  sf err %.2e   (|out| max %.2f)" % (e(ref32.double()), e(outs["f32"]), e(outs["sf"]), float(ref64.abs().max())))
