"""The input contract IN FRONT of the hot path, on the device:

  raw detector joints [B,17,2|3] (COCO order, pixels)  --add pelvis/neck-->  [B,19,..]      demo/run.py:103-121, data/PW3D/dataset.py:168-183
  --bbox, affine to 288x384, /[288,384], per-axis standardise-->  pose2d [B,J,2]          demo/run.py:127-134, data/PW3D/dataset.py:241-250

With rot = 0 / flip = 0 (all evaluation paths) the bbox/affine//[W,H] part is a per-axis positive scale + shift and cancels in the
standardisation, so the kernel computes (xy - mean) / std over the sample's joints (population std) -- checked against the
reference's own chain on the demo input (tests/golden/demo_preprocess.npz)."""
import ctypes

import torch

from . import _lib


def normalise_pose2d(joints, add_pelvis_neck=False):
    """joints [B,J,2|3] f32 on a HIP device (pixels) -> pose2d [B,J(+2),2] f32, the input of GATOR.forward."""
    if not joints.is_cuda:
        raise RuntimeError('normalise_pose2d: joints must live on a HIP device (there is no CPU path)')
    if joints.dim() != 3 or joints.shape[2] < 2:
        raise ValueError('normalise_pose2d: expected [B,J,2|3], got %s' % (tuple(joints.shape),))
    x = joints.contiguous().float()
    B, J, C = x.shape
    out = torch.empty((B, J + (2 if add_pelvis_neck else 0), 2), device=x.device, dtype=torch.float32)
    st = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(_lib.load().gator_preprocess_pose2d_f32(x.data_ptr(), B, J, C, int(bool(add_pelvis_neck)), out.data_ptr(), st),
               'gator_preprocess_pose2d_f32')
    return out


def coco_to_model_input(joints17):
    """COCO detector output [B,17,2|3] -> the 19-joint model input [B,19,2] (pelvis and neck appended, then standardised)."""
    if joints17.shape[1] != 17:
        raise ValueError('coco_to_model_input: expected 17 COCO joints, got %d' % joints17.shape[1])
    return normalise_pose2d(joints17, add_pelvis_neck=True)


def preprocess_chain(joints, rot_deg=None, flip=None, flip_pairs=(), add_pelvis_neck=False, res=(288, 384)):
    """The reference's whole input chain on the device (gator_preprocess_chain_f32): bbox -> process_bbox -> affine with rotation /
    flip -> /[W,H] -> standardise.  joints [B,J,2|3] f32 (pixels); rot_deg [B] f32 or None; flip [B] int32 or None.
    -> (pose2d [B,J(+2),2], valid [B] int32: 0 where process_bbox rejects the box and the reference drops the sample)."""
    if not joints.is_cuda:
        raise RuntimeError('preprocess_chain: joints must live on a HIP device (there is no CPU path)')
    x = joints.contiguous().float()
    B, J, C = x.shape
    dev = x.device
    out = torch.empty((B, J + (2 if add_pelvis_neck else 0), 2), device=dev, dtype=torch.float32)
    valid = torch.empty((B,), device=dev, dtype=torch.int32)
    rot = None if rot_deg is None else torch.as_tensor(rot_deg, dtype=torch.float32, device=dev).contiguous()
    fl = None if flip is None else torch.as_tensor(flip, dtype=torch.int32, device=dev).contiguous()
    pairs = torch.as_tensor(list(flip_pairs), dtype=torch.int32, device=dev).reshape(-1, 2).contiguous()
    st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(_lib.load().gator_preprocess_chain_f32(x.data_ptr(), B, J, C, int(bool(add_pelvis_neck)), rot.data_ptr() if rot is not None else None,
                                                      fl.data_ptr() if fl is not None else None, pairs.data_ptr() if pairs.numel() else None,
                                                      int(pairs.shape[0]), int(res[0]), int(res[1]), out.data_ptr(), valid.data_ptr(), st),
               'gator_preprocess_chain_f32')
    return out, valid
