import sys, time
sys.path.insert(0, '.')
import libjxl_amd as J
img = J.synth_image(3840, 2160, 177)
ctx = J.HipContext()
for i in range(3):
    t = {}
    t0 = time.perf_counter()
    d = J.encode_rgb8_gpu(img, ctx, timings=t, device_tokens=True, distance=1.0, cfl_fit=1)
    sys.stderr.write("[py] total %.4f %s now %.4f\n" % (time.perf_counter() - t0, t, time.monotonic()))
