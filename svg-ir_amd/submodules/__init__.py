"""Drop-in counterparts of the reference's `submodules` packages that sit on the hot path."""
