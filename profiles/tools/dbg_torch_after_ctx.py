import sys; sys.path.insert(0, '.')
import numpy as np
from ionotomo_amd import _lib
c = _lib.Context(0)
print("ctx ok")
import torch
print("torch available (ctx alive):", torch.cuda.is_available(), torch.cuda.device_count())
c.close()
print("after close:", torch.cuda.is_available())
