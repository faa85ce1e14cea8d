"""Physical constants of the PGW computation (values of the reference's constants.py:3-7,
which cites COSMO data_constants.f90)."""
CON_RD = 287.05       # gas constant of dry air [J kg-1 K-1]
CON_G = 9.80665       # gravity [m s-2]
CON_MW_MD = 0.622     # molecular-mass ratio water / dry air
