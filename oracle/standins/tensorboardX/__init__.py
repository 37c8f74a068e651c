"""Inert import stand-in (TEST INFRASTRUCTURE): run/logger.py imports SummaryWriter at module scope."""


class SummaryWriter:
    def __init__(self, *a, **k):
        raise RuntimeError('tensorboardX stand-in: logging is not available in the test environment')
