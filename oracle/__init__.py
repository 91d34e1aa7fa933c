"""CPU oracle (test infrastructure).  See oracle/oracle.cpp."""
