"""Configuration object with the attribute names the reference's DGCNN_FLAGS carries
(dgcnn/flags.py:9-45).  Only what the hot path reads is interpreted here (model.py:11-18,27;
trainval.py:17,26,31,47); the argparse CLI, IO and checkpoint flags are out of scope."""
from __future__ import annotations


class DGCNN_FLAGS(object):
    # flags for model (flags.py:9-17)
    NUM_CLASS = 2
    MODEL_NAME = "dgcnn"
    TRAIN = True
    KVALUE = 20
    DEBUG = False
    EDGE_CONV_LAYERS = 3
    EDGE_CONV_FILTERS = 64
    FC_LAYERS = 2
    FC_FILTERS = [512, 256]
    # flags for train/inference (flags.py:20-33)
    SEED = 1
    LEARNING_RATE = 0.001
    GPUS = [0]
    MINIBATCH_SIZE = 1
    NUM_POINT = 2048
    NUM_CHANNEL = 3
    BATCH_SIZE = 1
    WEIGHT_KEY = ""

    def __init__(self, **kw):
        self.update(kw)

    def update(self, args):
        """flags.py:150-167: upper-case the keys, split comma lists."""
        for name, value in args.items():
            setattr(self, name.upper(), value)
        for key in ("EDGE_CONV_FILTERS", "FC_FILTERS"):
            v = getattr(self, key)
            if isinstance(v, str):
                setattr(self, key, [int(a) for a in v.split(",")] if "," in v else int(v))
        if isinstance(self.GPUS, str):
            self.GPUS = [int(g) for g in self.GPUS.split(",")]
        return self
