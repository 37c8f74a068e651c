"""Inert import stand-in (TEST INFRASTRUCTURE): datasets/phoregen.py imports lmdb at module scope for the training
dataset, which no test here opens."""


def open(*a, **k):
    raise RuntimeError('lmdb stand-in: no database access in the test environment')
