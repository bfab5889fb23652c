"""Input pipeline of the Cap2Det reader."""
