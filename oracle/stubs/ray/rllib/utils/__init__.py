def try_import_torch():
    import torch
    import torch.nn as nn
    return torch, nn
