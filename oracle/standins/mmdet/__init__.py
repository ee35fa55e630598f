"""Stand-in mmdet (test-only plumbing). See oracle/standins/README.md."""
