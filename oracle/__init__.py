"""CPU oracle (test infrastructure only -- see oracle.c).  Never imported by the product package."""
