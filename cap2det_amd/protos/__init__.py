"""Config schema + text-format reader mirroring the reference `protos/` package."""
