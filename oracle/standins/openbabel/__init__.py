from rdkit import _Inert

openbabel = _Inert()
