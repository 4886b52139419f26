# Physical constants used by the reference (SI). Values as in cora.util.units.
c = 299792458.0
t_sidereal = 23.9344696 * 3600.0
