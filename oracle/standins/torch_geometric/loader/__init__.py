class DataLoader:  # import-time name only
    def __init__(self, *a, **k):
        raise NotImplementedError


class DataListLoader:  # import-time name only (run/run.py:3)
    def __init__(self, *a, **k):
        raise NotImplementedError
