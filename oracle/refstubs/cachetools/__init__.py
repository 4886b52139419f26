class LRUCache(dict):
    def __init__(self, maxsize=0):
        dict.__init__(self)
