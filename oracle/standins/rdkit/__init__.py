"""Inert rdkit stub: import-time names only (post-processing is never executed by the oracle)."""


class _Inert:
    def __getattr__(self, k):
        return _Inert()

    def __call__(self, *a, **k):
        return _Inert()


Geometry = _Inert()
RDLogger = _Inert()
Chem = _Inert()
