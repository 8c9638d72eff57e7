"""Inert stand-in: the reference imports `upfirdn2d_cuda` at module import time
(imaginaire/third_party) but SceneDreamer never calls it."""


def __getattr__(name):
    raise NotImplementedError("upfirdn2d_cuda." + name + " is not on SceneDreamer's path and is not provided")
