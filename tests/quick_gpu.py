import sys, time
sys.path.insert(0,'gsplat.js_amd/py'); sys.path.insert(0,'.')
import numpy as np
import gsplat_hip as gh
from oracle import oracle as O
print(gh.load_library())
for name in ("C1","C3"):
    cfg = gh.synth.CONFIGS[name]
    rows = gh.synth.config_rows(name)
    data,pos = O.scene_pack(rows)
    r = gh.HIPRenderer(cfg['width'], cfg['height'], timing=True)
    print(r.device_info())
    r.set_raw_scene(data,pos)
    cam = gh.orbit_camera(5, width=cfg['width'], height=cfg['height'], fx=cfg['fx'])
    r.set_camera(cam)
    for it in range(3):
        r.render_async(); r.sync()
        print(name, r.stats())
    di = r.lastDepthIndex()
    v,p,vp = cam.f32()
    odi,okeys,omm = O.sort(vp,pos)
    print("sort mismatches", int((di!=odi).sum()))
    img = r.readPixelsFloat()
    oimg, _, V, D = O.render_scene(data,pos,v,p,vp,cam.fx,cam.fy,cfg['width'],cfg['height'],mode=1)
    print("V",V,"D",D,"img max err", np.abs(img-oimg).max(), "mean alpha", img[...,3].mean(), oimg[...,3].mean())
