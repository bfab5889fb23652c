"""Mirror of the reference `core/` names used on the hot path."""
