#-*- coding: utf-8 -*-

from .initialize import *
