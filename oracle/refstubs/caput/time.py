class Observer(object):
    def __init__(self, lon=0.0, lat=0.0, alt=0.0, **kwargs):
        self.longitude = lon
        self.latitude = lat
        self.altitude = alt
