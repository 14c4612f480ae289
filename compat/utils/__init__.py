"""Reference package `utils`."""
