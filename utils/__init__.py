"""Result tools with the command lines of the reference (merge_subdomains, compare_results)."""
