"""tools/repro_lost_pixels.py - 96x64x10spp three times over (init, two frames, cleanup): NaN pixels per frame and differences between the images.  The frame of six
workgroups that showed the chain pixel of an XCD without waves being lost (RT_XCD_QUEUES=1 before the leftovers rule); RT_CLEANUP_DEVICE_RESET=1 adds the reset."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cuda_raytracing_optimized_amd as rt
imgs = []
for k in range(3):
    sp, mt, cam = rt.scene_random_spheres(96, 64)
    fb = rt.initRendererSpheres(sp, mt, cam, 96, 64, 50)
    rt.runRenderer(10, 8, 8); a = np.array(fb, copy=True); rt.runRenderer(10, 8, 8)
    imgs.append(np.array(fb, copy=True)); rt.cleanupRenderer()
    print(k, "nan first/second frame", int(np.isnan(a).sum()), int(np.isnan(imgs[-1]).sum()), "differs from image 0:", int((imgs[0].view(np.uint32) != imgs[-1].view(np.uint32)).sum()),
          "first vs second", int((a.view(np.uint32) != imgs[-1].view(np.uint32)).sum()), flush=True)
    bad = np.argwhere(np.isnan(imgs[-1]).any(axis=2))
    if len(bad): print("  nan pixels (y, x):", bad[:12].tolist())
