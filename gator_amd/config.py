"""Minimal stand-in for the four `core.config.cfg` keys the reference's model files read
(lib/models/GAT.py:13,130, lib/models/MDR.py:16,115,162, lib/models/GATOR.py:13).  Unlike lib/core/config.py:26-39 it has
no import side effects (no experiment directories are created or wiped).  Constructor kwargs override it."""


class _NS(dict):
    __getattr__ = dict.__getitem__
    __setattr__ = dict.__setitem__


cfg = _NS()
cfg.DATASET = _NS(BASE_DATA_DIR='data/base_data')
cfg.MODEL = _NS(alpha=False, posenet_pretrained=False, posenet_path='')
