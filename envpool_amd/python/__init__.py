"""Host-side Python front-end (the counterpart of envpool2/python/)."""
