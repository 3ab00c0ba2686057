"""Reference-shaped plugin modules (llava.model.* counterparts) backed by libslime_hip."""
