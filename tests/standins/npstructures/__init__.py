"""TEST-ONLY stand-in for `npstructures` (HashTable / Counter import targets)."""
from .hashtable import HashTable, Counter  # noqa: F401
