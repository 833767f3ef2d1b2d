"""Drop-in package root for the reference `dynamicemb` package (MI355X-native)."""
