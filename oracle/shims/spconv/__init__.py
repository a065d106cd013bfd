"""Stand-in package so that `import spconv.pytorch as spconv` resolves -- see oracle/shims/README.md."""
