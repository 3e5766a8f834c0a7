"""ORACLE - test infrastructure only (see oracle/swe2d_oracle.py).  Never imported by thetis_amd."""
