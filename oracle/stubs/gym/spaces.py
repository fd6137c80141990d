import numpy as np


class Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        self.low = np.asarray(low)
        self.high = np.asarray(high)
        self.shape = tuple(self.low.shape) if shape is None else tuple(shape)
        self.dtype = np.dtype(dtype)
