"""Reference package `models.building_blocks`."""
