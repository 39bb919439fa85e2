"""Checkpoint loading with the reference's call shape and error convention (lib/funcs_utils.py:119-127): a checkpoint is a
`torch.save`d dict {'epoch', 'model_state_dict', 'optim_state_dict', 'scheduler_state_dict', 'train_log', 'test_log'}
(main/train.py:51-58); callers read only `model_state_dict` (lib/core/base.py:69-70, demo/run.py:97-98)."""
import torch


def load_checkpoint(load_dir, epoch=0, pick_best=False, map_location=None):
    """-> the checkpoint dict.  The reference maps to 'cuda'; here the default is the CPU (the module is moved afterwards with
    .cuda()/.to(), which is also when the device context is (re)built).  Missing / unreadable file -> ValueError, as upstream."""
    try:
        return torch.load(load_dir, map_location=map_location or 'cpu', weights_only=False)
    except Exception as e:      # noqa: BLE001  (the reference catches everything and re-raises ValueError)
        raise ValueError('No checkpoint exists!\n', e)
