"""Mirror of the reference `models/` plugin surface (builder / registry / ModelBase /
cap2det_model / label_extractor) for the MI355X hot path."""
