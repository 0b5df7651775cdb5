# -*- coding: utf-8 -*-
#
# pylint: disable=wildcard-import
from .pressure_correction import *    # noqa: F401,F403
