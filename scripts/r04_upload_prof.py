import sys, os
sys.path.insert(0, os.getcwd())
os.environ["JXLHIP_UPLOAD_PROF"] = "1"
import bench, libjxl_amd as J
data = bench.make_stream(3840, 2160, 1.0, 177)
f = J.Frame(data, threads=4)
c = J.HipContext(0)
for i in range(4):
    c.upload(f)
c.close(); f.close()
