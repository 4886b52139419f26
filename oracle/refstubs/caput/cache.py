class NumpyCache(dict):
    def __init__(self, size=0):
        dict.__init__(self)
