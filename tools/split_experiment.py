"""Experiment: does running the batch as N sub-batches on N HIP streams (separate contexts) beat one launch chain?"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
from gator_amd import synthetic
dev = torch.device('cuda')
B = 256
x = torch.from_numpy(synthetic.synthetic_pose2d(B, 17, 1)).to(dev)
def run(nsplit, offset_us=0, steps=30):
    models = [bench.build_model(17, 'fused', dev)[0] for _ in range(nsplit)]
    streams = [torch.cuda.Stream() for _ in range(nsplit)]
    sh = B // nsplit
    xs = [x[i * sh:(i + 1) * sh].contiguous() for i in range(nsplit)]
    def step():
        for i in range(nsplit):
            with torch.cuda.stream(streams[i]):
                models[i](xs[i])
    for _ in range(5): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print('nsplit %d: %.4f ms/step -> %.0f meshes/s' % (nsplit, dt * 1e3, B / dt))
for n in (1, 2, 4):
    run(n)
