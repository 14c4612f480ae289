"""Reference package `models`."""
