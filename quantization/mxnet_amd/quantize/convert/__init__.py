#-*- coding: utf-8
"""quantize.convert — same public names as the reference (quantize/convert/__init__.py:3-13)."""
from .convert_conv2d import *

from .convert_act import *

from .convert_bn import *

from .convert import *

from .convert_dense import *

from .ste_func import *

from . import wino_matrix
