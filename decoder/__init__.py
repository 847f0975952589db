"""Import surface of the reference (implicit namespace package there).  See INTEGRATION.md."""
