"""Import-only stand-in for `shapely` (oracle harness; lets RAiDER.models.weatherModel be imported in place)."""
