"""Mirror of the reference `train/` step semantics for the MI355X hot path."""
