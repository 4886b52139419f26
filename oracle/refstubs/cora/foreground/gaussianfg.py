"""cora sky models are not available in this container."""


class _NA(object):
    def __init__(self, *a, **k):
        raise NotImplementedError("cora sky models are not available here")


Corr21cm = EoR21cm = PointSources = FullSkySynchrotron = FullSkyPolarisedSynchrotron = _NA


def clarray(*a, **k):
    raise NotImplementedError("cora sky models are not available here")
