"""Gluon-shaped host facade (see block.py / parameter.py / nn.py)."""
from .parameter import Parameter, ParameterDict
from .block import Block, HybridBlock
from . import nn
from . import data
from . import model_zoo
from . import loss
from .trainer import Trainer
