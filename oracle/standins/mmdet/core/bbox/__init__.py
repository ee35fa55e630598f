class BaseSampler:
    def __init__(self, *args, **kwargs):
        pass


class SamplingResult:
    pass
