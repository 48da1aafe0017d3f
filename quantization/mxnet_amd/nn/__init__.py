#-*- coding: utf-8 -*-

from .quantized_conv import *
