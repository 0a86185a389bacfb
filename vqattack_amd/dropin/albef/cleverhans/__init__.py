"""Drop-in `cleverhans` package (albef flavor) backed by vqattack_amd -- see vqattack_amd/dropin/__init__.py."""
