"""Import-path compatibility with the reference package layout (re-exports only)."""
