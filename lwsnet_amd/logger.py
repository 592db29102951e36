"""Logging helpers with the behaviour of /root/reference/utils/logger.py:7-41 and utils/utils.py:1-17.

`setup_logger(name, save_path)` logs to stderr and, when `save_path` is given, to `<save_path>/<name>-<Y-m-d-H-M>.log`
with the reference's line format.  One deliberate difference: the reference returns None when the same logger is set up
twice (logger.py:25-26, its callers never do); here the existing logger is returned.
"""
from __future__ import annotations

import logging
import os
import sys
import time

FORMAT = "[%(asctime)s %(filename)s:%(lineno)s] %(levelname)s: %(message)s"
DATEFMT = "%Y-%m-%d %H:%M:%S"


def log_file_name(name, now=None):
    """`train.py` -> `train_py-2021-01-31-12-00.log`; a name with a directory part keeps only the directory (logger.py:8-13)."""
    stamp = time.strftime("-%Y-%m-%d-%H-%M", time.localtime(time.time() if now is None else now))
    name = name.replace(".", "_")
    head = os.path.dirname(name)
    return (name if head == "" else head) + stamp + ".log"


def setup_logger(name, save_path=None, now=None):
    file_name = log_file_name(name, now)
    logger = logging.getLogger(file_name)
    if getattr(logger, "_lws_ready", False):
        return logger
    fmt = logging.Formatter(FORMAT, datefmt=DATEFMT)
    stream = logging.StreamHandler(stream=sys.stderr)
    stream.setLevel(logging.DEBUG)
    stream.setFormatter(fmt)
    logger.addHandler(stream)
    if save_path is not None:
        os.makedirs(save_path, exist_ok=True)
        fh = logging.FileHandler(os.path.join(save_path, file_name))
        fh.set_name(file_name)
        fh.setLevel(logging.DEBUG)
        fh.setFormatter(fmt)
        logger.addHandler(fh)
    logger.setLevel(logging.DEBUG)
    logger._lws_ready = True
    return logger


class AverageMeter:
    """Running average (utils/utils.py:1-17)."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count
