# -*- coding: utf-8 -*-
from . import _hip  # noqa: F401
