"""Binary-fluid (Shan-Chen) simulation scripts."""
