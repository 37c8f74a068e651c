class DataLoader:  # import-time name only
    def __init__(self, *a, **k):
        raise NotImplementedError
