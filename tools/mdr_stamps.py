import sys, torch
sys.path.insert(0, '.')
import bench
m, base, alpha = bench.build_model(17, 'fused', torch.device('cuda'))
from gator_amd import synthetic
x = torch.from_numpy(synthetic.synthetic_pose2d(256, 17, 1)).cuda()
for i in range(2):
    m(x); torch.cuda.synchronize()
