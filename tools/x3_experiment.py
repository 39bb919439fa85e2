"""Experiment: split-precision (bf16x3) vertex regressor vs the fp32-MFMA one: error against fp64 and kernel time.
Run twice: with and without GATOR_UPSAMPLE_X3=1 (the switch is read once per process)."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.helpers import build_model

def main():
    B = int(os.environ.get('B', 256))
    model = build_model('h36m17_bn')[1].eval()
    sd = model.state_dict()
    g = torch.Generator().manual_seed(1)
    vc = (torch.randn(B, 431, 3, generator=g) * 0.3).cuda()
    w = sd['pose2mesh.upsample_conv.weight'].double().cuda(); b = sd['pose2mesh.upsample_conv.bias'].double().cuda()
    tpl = model.pose2mesh.template_6890.double().cuda() if hasattr(model.pose2mesh, 'template_6890') else None
    ref = torch.nn.functional.conv1d(vc.double(), w, b, padding=1)
    model.pose2mesh.upsample(vc); torch.cuda.synchronize()
    if tpl is None:
        z = model.pose2mesh.upsample(torch.zeros_like(vc)).double()      # bias + template, exact to fp32
        ref = ref - b[None, :, None] + z
    else:
        ref = ref + tpl[None]
    model.pose2mesh.profile(1)
    for _ in range(3): out = model.pose2mesh.upsample(vc)
    torch.cuda.synchronize()
    model.pose2mesh.profile_read()
    for _ in range(20): out = model.pose2mesh.upsample(vc)
    torch.cuda.synchronize()
    pr = model.pose2mesh.profile_read()
    err = (out.double() - ref).abs()
    print(json.dumps({'x3': bool(os.environ.get('GATOR_UPSAMPLE_X3')), 'B': B, 'max_err': err.max().item(), 'rms_err': err.pow(2).mean().sqrt().item(),
                      'ref_absmax': ref.abs().max().item(), 'stages_ms': pr}))

if __name__ == '__main__':
    main()
